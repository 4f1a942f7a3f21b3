// The forward of the LONG tiles of a big one-view launch (seg_mode 2: one view of more than 8,192 tiles - the texture pass -, tiles of
// at least kSegLongMin pairs), parallel along DEPTH.
//
// k_render_fwd walks a tile's list front to back in one workgroup, 3-6 us per 128 positions: a list of 12,614 pairs lasts 300 us - as
// long as the whole rest of a 10^6-Gaussian view (the same wave-steps take the forward 158 us on a scene whose longest list holds
// 1,459 pairs, 299 / 443 us from two cameras of one with lists of 9,000 - 12,000).  The blend is a linear recurrence in the state
// (T, C, D) - T' = T (1 - a), C' = C + c a T - so a SEGMENT of the list acts on the state in front of it as
//     T_out = T_in T_seg,   C_out = C_in + T_in C_seg,   D_out = D_in + T_in D_seg
// with (T_seg, C_seg, D_seg) the segment blended from T = 1.  What is not linear is the stop rule (stop before the splat that would
// take T below 1e-4): but T only decreases, so a pixel stops inside the FIRST segment at whose END T_in T_seg falls below the
// threshold, and nowhere before.  Three launches over the slots the segmented backward owns anyway (kSeg = 128 positions each):
//   1. k_fwd_long_seg<false>  one workgroup per segment: blends it from T = 1 and keeps (T_seg, C_seg, D_seg, last contributor) in
//      the segment's snapshot slot;
//   2. k_fwd_long_prefix      one workgroup per long tile, one thread per pixel, no cooperation: multiplies the prefix through the
//      segments, overwrites each slot with what the backward wants there (the state in front of the next boundary), finds the
//      segment a pixel stops in - or finishes the pixel if it never stops;
//   3. k_fwd_long_seg<true>   one workgroup per segment in which some pixel stops: walks it again from that pixel's true state with
//      the stop rule, and finishes those pixels.
// The sums are associated differently from the one-pass forward's (T_in C_seg against c a T splat by splat: ~1e-7 relative), and a
// pixel whose transmittance comes within rounding of the threshold at a segment's end may stop one contributing splat earlier or
// later than the one-pass walk would - the class of the alpha >= 1/255 threshold pixels (HISTORY.md section 2), weight <= 1e-4.
#pragma once

constexpr uint32_t kStopCode = 0x80000000u;          // n_contrib between launches 2 and 3: kStopCode | the segment the pixel stops in

template <bool FINISH>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_fwd_long_seg(const KP kp)
{
    constexpr int kS = kSeg, kChunks = kS / 64, kU = 4;
    constexpr int kNull = kS;
    constexpr int kListStride = kS + 8;
    constexpr int kRec = 48;
    static_assert(kS <= kBlock && kS % 64 == 0, "one staging thread per list position of a segment");
    __shared__ __attribute__((aligned(16))) unsigned char s_rec[(kS + 1) * kRec];
    __shared__ __attribute__((aligned(8))) unsigned short s_list[4][4][kListStride];
    __shared__ uint32_t s_any[4];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, row = lane >> 4;
    if (tid < kRec / 4) reinterpret_cast<float *>(s_rec + kNull * kRec)[tid] = 0.f;
    const uint32_t n_live = kp.status->live_segments;        // (the compact list of the segments: write_segment_slots)
    for (uint32_t k = blockIdx.x; k < n_live; k += kp.tile_blocks) {
        const uint4 it = kp.slot_tab[kp.live[k]];
        const uint32_t j = it.w & 0x7fffffffu;
        const int v = (int)(it.x >> 20), t_ = (int)(it.x & 0xfffffu);
        const int ty = t_ / kp.gx, tx = t_ - ty * kp.gx;
        const uint32_t off = it.y, n = it.z;
        const uint32_t lo = j * (uint32_t)kS;
        const uint32_t nb = (n + (uint32_t)kS - 1u) / (uint32_t)kS;
        int px, py;
        tile_pixel(tid, tx, ty, px, py);
        const bool inside = px < kp.W && py < kp.H;
        const v2f pix_f = { (float)px, (float)py };
        const size_t HW = (size_t)kp.H * kp.W, pix = (size_t)py * kp.W + px;
        float *slot0 = kp.snap + ((size_t)v * kp.slots_per_view + seg_slot0(kp, off, (uint32_t)t_)) * (kSnapFloats * kBlock) + tid;
        float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f;
        uint32_t last_contributor = 0u;
        bool mine = inside;                          // does this pixel take splats in this launch?
        if (FINISH) {
            mine = inside && kp.n_contrib[(size_t)v * HW + pix] == (kStopCode | j);
            const unsigned long long any = __ballot(mine);
            if (lane == 0) s_any[wave] = any != 0ull ? 1u : 0u;
            __syncthreads();
            const bool some = (s_any[0] | s_any[1] | s_any[2] | s_any[3]) != 0u;
            __syncthreads();                         // (s_any is rewritten by the next item)
            if (!some) continue;                     // workgroup-uniform: nobody stops in this segment
            if (mine && j != 0u) {                   // the state in front of this segment: launch 2 left it in the slot before
                const float *sp = slot0 + (size_t)(j - 1u) * (kSnapFloats * kBlock);
                T = sp[0]; C0 = sp[kBlock]; C1 = sp[2 * kBlock]; C2 = sp[3 * kBlock]; D = sp[4 * kBlock];
                last_contributor = __float_as_uint(sp[5 * kBlock]);
            }
        }
        unsigned long long done_m = __ballot(!mine);
        // ---- stage the segment's splats (k_render_fwd's throughput staging)
        if (tid < kS) {
            float4 head = make_float4(0.f, 0.f, -1.f, 0.f);      // (x, y, cut-off r2, -): a slot without a splat touches nothing
            if (lo + tid < n) {
                const unsigned long long key = kp.keys[(size_t)v * kp.cap + off + lo + tid];
                const uint32_t g = (uint32_t)key;
                if (g < (uint32_t)kp.P) {                        // (stale entries of a truncated list are ignored)
                    const float2 p = kp.xy[(size_t)v * kp.P + g];
                    const float4 c = kp.conic_opacity[(size_t)v * kp.P + g];
                    const float *rgb = kp.shs ? kp.rgb + (size_t)v * kp.P * 3 : kp.colors_precomp + 3 * param_row0(kp, v);
                    unsigned char *rec = s_rec + tid * kRec;
                    head = make_float4(p.x, p.y, cutoff_radius2(c), 0.f);
                    *reinterpret_cast<float4 *>(rec + 16) = scale_conic(c);
                    *reinterpret_cast<float4 *>(rec + 32) = make_float4(rgb[3 * (size_t)g], rgb[3 * (size_t)g + 1], rgb[3 * (size_t)g + 2],
                                                                       __uint_as_float((uint32_t)(key >> 32)));
                }
                if (!FINISH) kp.cut_r2[(size_t)v * kp.cap + off + lo + tid] = head.z;     // the backward stages the same splats: it reads the cut-off back
            }
            *reinterpret_cast<float4 *>(s_rec + tid * kRec) = head;
        }
        __syncthreads();
        uint32_t last_e = 0xffffffffu;
        if (done_m != ~0ull) {                       // wave-uniform
            // which of the staged splats can touch which of this wave's four sub-blocks (= DPP rows)
            // (launch 3: a sub-block none of whose pixels stops here walks an empty list - most of them)
            uint32_t rows_done = 0u;
            if (FINISH) {
#pragma unroll
                for (int r = 0; r < 4; r++) rows_done |= (((done_m >> (16 * r)) & 0xffffull) == 0xffffull ? 1u : 0u) << r;
            }
            unsigned long long m[4][kChunks];
#pragma unroll
            for (int c4 = 0; c4 < kChunks; c4++) {
                unsigned long long mc[4] = { 0ull, 0ull, 0ull, 0ull };
                if (lo + ((uint32_t)c4 << 6) < n) {
                    const float4 head = *reinterpret_cast<const float4 *>(s_rec + ((c4 << 6) + lane) * kRec);
                    wave_touch_masks(make_float2(head.x, head.y), head.z, tx, ty, wave, mc);
                    if (FINISH && rows_done != 0u) {
#pragma unroll
                        for (int r = 0; r < 4; r++) mc[r] = ((rows_done >> r) & 1u) ? 0ull : mc[r];
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; r++) m[r][c4] = mc[r];
            }
            int nsteps = 0, cnts[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                cnts[r] = build_visit_list<kChunks, false, kRec>(m[r], s_list[wave][r], lane, 0);
                nsteps = max(nsteps, cnts[r]);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) pad_visit_list<kU>(s_list[wave][r], cnts[r], nsteps, lane, (unsigned short)(kNull * kRec));
            __builtin_amdgcn_wave_barrier();
            const unsigned short *list = s_list[wave][row];
            for (int k = 0; k < nsteps; k += kU) {
                uint32_t e[kU];
                const uint2 pk = *reinterpret_cast<const uint2 *>(list + k);
                e[0] = pk.x & 0xffffu; e[1] = pk.x >> 16; e[2] = pk.y & 0xffffu; e[3] = pk.y >> 16;
                float alpha[kU];
                unsigned long long valid[kU];
#pragma unroll
                for (int u = 0; u < kU; u++) {
                    const v2f g_xy = *reinterpret_cast<const v2f *>(s_rec + e[u]);
                    float p2, G;
                    eval_splat(*reinterpret_cast<const float4 *>(s_rec + e[u] + 16), g_xy - pix_f, p2, G, alpha[u]);
                    valid[u] = __ballot(!(p2 > 0.0f)) & __ballot(!(alpha[u] < T4D_ALPHA_MIN));
                }
#pragma unroll
                for (int u = 0; u < kU; u++) {       // blending is sequential in list order (k_render_fwd's throughput blend)
                    const float test_T = T * (1.f - alpha[u]);
                    const unsigned long long below = __ballot(test_T < T4D_T_STOP);
                    const unsigned long long live = valid[u] & ~done_m;
                    done_m |= live & below;
                    const bool ok = __builtin_amdgcn_inverse_ballot_w64(live & ~below);
                    const float4 cd = *reinterpret_cast<const float4 *>(s_rec + e[u] + 32);
                    const float w = ok ? alpha[u] * T : 0.f;
                    C0 = fmaf(cd.x, w, C0); C1 = fmaf(cd.y, w, C1); C2 = fmaf(cd.z, w, C2);
                    D = fmaf(cd.w, w, D);
                    T = ok ? test_T : T;
                    last_e = ok ? e[u] : last_e;
                }
                if (done_m == ~0ull) break;
            }
        }
        if (last_e != 0xffffffffu) last_contributor = lo + ((last_e * 43691u) >> 21) + 1u;     // entry / 48 for entries < 2^17
        if (!FINISH) {
            // the segment as an operator on the state in front of it; a negative T: the segment stops a pixel even from T = 1
            float *sp = slot0 + (size_t)j * (kSnapFloats * kBlock);
            const bool stopped = inside && ((done_m >> lane) & 1ull) != 0ull;
            sp[0] = stopped ? -T : T; sp[kBlock] = C0; sp[2 * kBlock] = C1; sp[3 * kBlock] = C2; sp[4 * kBlock] = D;
            sp[5 * kBlock] = __uint_as_float(last_contributor);
        } else if (mine) {
            // the pixel is finished (by the stop rule inside this segment - or, within rounding of the threshold, at its end)
            const float *vr = kp.views + (size_t)v * T4D_VIEW_FLOATS;
            kp.final_T[(size_t)v * HW + pix] = T;
            kp.n_contrib[(size_t)v * HW + pix] = last_contributor;
            float *oc = kp.out_color + (size_t)v * 3 * HW;
            oc[pix] = C0 + T * vr[35];
            oc[HW + pix] = C1 + T * vr[36];
            oc[2 * HW + pix] = C2 + T * vr[37];
            kp.out_depth[(size_t)v * HW + pix] = D;
            kp.out_alpha[(size_t)v * HW + pix] = 1.f - T;
            float *sf = slot0 + (size_t)(nb - 1u) * (kSnapFloats * kBlock);       // the final state, in the tile's last slot
            sf[0] = T; sf[kBlock] = C0; sf[2 * kBlock] = C1; sf[3 * kBlock] = C2; sf[4 * kBlock] = D;
        }
        __syncthreads();                             // staging buffers are reused by the next item
    }
}

// Launch 2: one workgroup per long tile (the length-ordered work items begin with them), one thread per pixel.
__global__ __launch_bounds__(kBlock) void k_fwd_long_prefix(const KP kp)
{
    constexpr int kS = kSeg;
    const int tid = threadIdx.x;
    for (uint32_t item = blockIdx.x; item < (uint32_t)(kp.V * kp.T); item += gridDim.x) {
        const uint4 it = kp.items[item];
        const uint32_t off = it.y, n = it.z;
        if (n < kp.seg_min_pairs) break;             // ordered by length class: ...
        const int v = (int)(it.x >> 20), t_ = (int)(it.x & 0xfffffu);
        if (kp.slot_tab[(size_t)v * kp.slots_per_view + seg_slot0(kp, off, (uint32_t)t_)].w == 0u) continue;      // (... a class may hold shorter tiles too)
        const int ty = t_ / kp.gx, tx = t_ - ty * kp.gx;
        int px, py;
        tile_pixel(tid, tx, ty, px, py);
        const bool inside = px < kp.W && py < kp.H;
        const size_t HW = (size_t)kp.H * kp.W, pix = (size_t)py * kp.W + px;
        const uint32_t nb = (n + (uint32_t)kS - 1u) / (uint32_t)kS;
        float *slot0 = kp.snap + ((size_t)v * kp.slots_per_view + seg_slot0(kp, off, (uint32_t)t_)) * (kSnapFloats * kBlock) + tid;
        float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f;
        uint32_t last = 0u, stop_seg = 0xffffffffu;
        // Four segments' records per round, and the NEXT round's requested before this round's are used and overwritten (loads and
        // stores share one counter on this chip: requested behind the stores, every round waited for its own stores first - a
        // memory round trip per segment, 26 us for a list of 99).
        constexpr int kAhead = 4;
        float qn[kAhead][6];
        auto request = [&](const uint32_t j0) {
#pragma unroll
            for (int a = 0; a < kAhead; a++) {
                const float *sn = slot0 + (size_t)min(j0 + (uint32_t)a, nb - 1u) * (kSnapFloats * kBlock);
#pragma unroll
                for (int f = 0; f < 6; f++) qn[a][f] = sn[f * kBlock];
            }
        };
        request(0u);
        bool wave_done = false;
        for (uint32_t j0 = 0; j0 < nb && !wave_done; j0 += kAhead) {
            float q[kAhead][6];
#pragma unroll
            for (int a = 0; a < kAhead; a++)
#pragma unroll
                for (int f = 0; f < 6; f++) q[a][f] = qn[a][f];
            if (j0 + kAhead < nb) request(j0 + kAhead);
#pragma unroll
            for (int a = 0; a < kAhead; a++) {
                const uint32_t j = j0 + (uint32_t)a;
                if (j >= nb) break;                  // workgroup-uniform
                float *sp = slot0 + (size_t)j * (kSnapFloats * kBlock);
                if (stop_seg == 0xffffffffu) {
                    const float t_out = T * fabsf(q[a][0]);
                    if (q[a][0] < 0.f || t_out < T4D_T_STOP) {
                        stop_seg = j;                // launch 3 walks this segment from the state in the slot before
                    } else {
                        C0 = fmaf(T, q[a][1], C0); C1 = fmaf(T, q[a][2], C1); C2 = fmaf(T, q[a][3], C2); D = fmaf(T, q[a][4], D);
                        const uint32_t lc = __float_as_uint(q[a][5]);
                        last = lc != 0u ? lc : last;
                        T = t_out;
                        // what the backward reads at the boundary behind this segment (and launch 3 as the state in front of the next)
                        sp[0] = T; sp[kBlock] = C0; sp[2 * kBlock] = C1; sp[3 * kBlock] = C2; sp[4 * kBlock] = D;
                        sp[5 * kBlock] = __uint_as_float(last);
                    }
                }
            }
            wave_done = __ballot(stop_seg == 0xffffffffu) == 0ull;          // the wave's pixels have all found their segment
        }
        if (!inside) continue;
        if (stop_seg != 0xffffffffu) {
            kp.n_contrib[(size_t)v * HW + pix] = kStopCode | stop_seg;
            continue;
        }
        // never stopped: the state behind the last segment is the pixel's (it sits in the last slot already)
        const float *vr = kp.views + (size_t)v * T4D_VIEW_FLOATS;
        kp.final_T[(size_t)v * HW + pix] = T;
        kp.n_contrib[(size_t)v * HW + pix] = last;
        float *oc = kp.out_color + (size_t)v * 3 * HW;
        oc[pix] = C0 + T * vr[35];
        oc[HW + pix] = C1 + T * vr[36];
        oc[2 * HW + pix] = C2 + T * vr[37];
        kp.out_depth[(size_t)v * HW + pix] = D;
        kp.out_alpha[(size_t)v * HW + pix] = 1.f - T;
    }
}
