// t4d_raster_binning.h - part of the translation unit t4d_raster.hip (included there, inside its anonymous namespace; not a
// stand-alone header).  A.1 / A.2: preprocess (cull, project, EWA, tile counting, pair slots), per-view tile scan, scatter into tile bins, the one-launch variants for a single small view.
// See t4d_raster.hip for the overview, the constants, the state layout and the kernel parameter block.
// ---------------------------------------------------------------------------------------------------------
// A.1 preprocess (+ tile counting + pair-slot allocation)
// ---------------------------------------------------------------------------------------------------------
// what a thread of the preprocess pass knows about its Gaussian afterwards (k_front_small goes on from here without re-reading it)
struct PreOut {
    uint32_t tiles, pbase;          // tiles touched; first pair slot (valid when fits)
    int x0, y0, x1, y1;             // tile rectangle
    float depth;
    bool fits;
};

__device__ __forceinline__ void preprocess_body(const KP &kp, const uint32_t gb, const uint32_t vb, PreOut &po)
{
#pragma clang fp contract(off)
    __shared__ uint32_t s_wave_tot[4];
    __shared__ uint32_t s_base;
    __shared__ int s_bb[4];
    __shared__ uint32_t s_hist[kHist], s_hbase[kHist];
    const int tid = threadIdx.x;
    const int g = (int)gb * kBlock + tid;
    const int v = (int)vb;
    po.tiles = 0; po.pbase = 0; po.x0 = po.y0 = po.x1 = po.y1 = 0; po.depth = 0.f; po.fits = false;
    const ViewRecord vrec = load_view_record(kp.views, v);
    const float *view = vrec.view, *proj = vrec.proj;
    const size_t vg = (size_t)v * kp.P + g;
    const size_t gp = param_row0(kp, v) + (size_t)g;          // row of this Gaussian in its view's parameter set

    uint32_t tiles = 0;
    int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    if (g < kp.P) {
        const float mean[3] = { kp.means3D[3 * gp], kp.means3D[3 * gp + 1], kp.means3D[3 * gp + 2] };
        // (covariance parameters and opacity are requested together with the mean, not behind the near-plane test)
        float cov3_in[6] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f }, sc[3] = { 0.f, 0.f, 0.f };
        float4 q = make_float4(1.f, 0.f, 0.f, 0.f);
        if (kp.cov3D_precomp) {
#pragma unroll
            for (int k = 0; k < 6; k++) cov3_in[k] = kp.cov3D_precomp[6 * gp + k];
        } else {
            sc[0] = kp.scales[3 * gp]; sc[1] = kp.scales[3 * gp + 1]; sc[2] = kp.scales[3 * gp + 2];
            q = reinterpret_cast<const float4 *>(kp.rotations)[gp];
        }
        float opacity = kp.opacities[gp];
        if (kp.raw_params) {                             // T4D_FLAG_RAW_PARAMS: the optimiser's parameters (helpers.py:95-97)
            opacity = t4d_act_sigmoid(opacity);
            if (!kp.cov3D_precomp) {
                sc[0] = t4d_act_exp(sc[0]); sc[1] = t4d_act_exp(sc[1]); sc[2] = t4d_act_exp(sc[2]);
                q = t4d_act_normalize(q);
            }
        }
        int radius = 0;
        const float pvz = view[2] * mean[0] + view[6] * mean[1] + view[10] * mean[2] + view[14];
        if (pvz > T4D_NEAR_CULL_Z) {
            const float hx = proj[0] * mean[0] + proj[4] * mean[1] + proj[8] * mean[2] + proj[12];
            const float hy = proj[1] * mean[0] + proj[5] * mean[1] + proj[9] * mean[2] + proj[13];
            const float hw = proj[3] * mean[0] + proj[7] * mean[1] + proj[11] * mean[2] + proj[15];
            const float pw = 1.0f / (hw + T4D_HOM_W_EPS);
            float cov3[6];
            if (kp.cov3D_precomp) {
#pragma unroll
                for (int k = 0; k < 6; k++) cov3[k] = cov3_in[k];
            } else {
                cov3d_from_scale_rot(sc, kp.scale_modifier, q, cov3);
            }
            const float tanx = vrec.tanx, tany = vrec.tany;
            const float fx = kp.W / (2.0f * tanx), fy = kp.H / (2.0f * tany);
            float T0[3], T1[3], t[3];
            bool inx, iny;
            ewa_rows(mean, view, fx, fy, tanx, tany, T0, T1, t, inx, iny);
            float v0[3], v1[3];
            sym3_mul(cov3, T0, v0);
            sym3_mul(cov3, T1, v1);
            const float a = T0[0] * v0[0] + T0[1] * v0[1] + T0[2] * v0[2] + T4D_COV2D_DILATION;
            const float b = T0[0] * v1[0] + T0[1] * v1[1] + T0[2] * v1[2];
            const float c = T1[0] * v1[0] + T1[1] * v1[1] + T1[2] * v1[2] + T4D_COV2D_DILATION;
            const float det = a * c - b * b;
            if (det != 0.0f) {
                const float det_inv = 1.f / det;
                const float mid = 0.5f * (a + c);
                const float l1 = mid + sqrtf(fmaxf(T4D_EIGEN_FLOOR, mid * mid - det));
                const float l2 = mid - sqrtf(fmaxf(T4D_EIGEN_FLOOR, mid * mid - det));
                const float my_radius = ceilf(T4D_RADIUS_SIGMAS * sqrtf(fmaxf(l1, l2)));
                const float px = ndc2pix(hx * pw, kp.W), py = ndc2pix(hy * pw, kp.H);
                tile_rect(px, py, (int)my_radius, kp.gx, kp.gy, x0, y0, x1, y1);
                tiles = (uint32_t)((x1 - x0) * (y1 - y0));
                if (tiles > 0) {
                    radius = (int)my_radius;
                    po.depth = pvz;
                    kp.xy[vg] = make_float2(px, py);
                    kp.depth[vg] = pvz;
                    kp.conic_opacity[vg] = make_float4(c * det_inv, -b * det_inv, a * det_inv, opacity);
                    if (kp.shs) {
                        float d[3] = { mean[0] - vrec.campos[0], mean[1] - vrec.campos[1], mean[2] - vrec.campos[2] };
                        const float len = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
                        d[0] /= len; d[1] /= len; d[2] /= len;
                        float bas[16];
                        sh_basis(kp.deg, d, bas);
                        const int K = (kp.deg + 1) * (kp.deg + 1);
                        const float *sh = kp.shs + gp * kp.M * 3;
                        // a Gaussian's coefficients are 12*M contiguous bytes: fetch them as 16-byte loads when the row
                        // is 16-byte aligned (M % 4 == 0, e.g. the 16 coefficients of degree 3) instead of 3*K scalar
                        // loads at a 12*M-byte lane stride
                        float shl[48];
                        if ((kp.M & 3) == 0 && kp.M <= 16) {
                            const float4 *sh4 = reinterpret_cast<const float4 *>(sh);
#pragma unroll
                            for (int i = 0; i < 12; i++)
                                if (i * 4 < K * 3) {
                                    const float4 t4 = sh4[i];
                                    shl[4 * i] = t4.x; shl[4 * i + 1] = t4.y; shl[4 * i + 2] = t4.z; shl[4 * i + 3] = t4.w;
                                }
                        } else {
#pragma unroll
                            for (int i = 0; i < 48; i++)
                                if (i < K * 3) shl[i] = sh[i];
                        }
                        uint32_t cl = 0;
#pragma unroll
                        for (int ch = 0; ch < 3; ch++) {
                            float r = 0.f;
#pragma unroll
                            for (int k = 0; k < 16; k++)
                                if (k < K) r += bas[k] * shl[k * 3 + ch];
                            r += 0.5f;
                            if (r < 0.f) cl |= 1u << ch;
                            kp.rgb[vg * 3 + ch] = fmaxf(r, 0.f);
                        }
                        kp.clamped[vg] = (uint8_t)cl;
                    }
                }
            }
        }
        kp.radii[vg] = radius;
    }

    // ---- pair slots: block-local exclusive scan, ONE returning atomic per workgroup on the view's cursor ----
    const uint32_t incl = wave_incl_scan(tiles);
    const int wave = tid >> 6, lane = tid & 63;
    if (lane == 63) s_wave_tot[wave] = incl;
    if (tid == 0) { s_bb[0] = 0x7fffffff; s_bb[1] = 0x7fffffff; s_bb[2] = 0; s_bb[3] = 0; }
    __syncthreads();
    uint32_t wave_off = 0, block_tot = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const uint32_t t = s_wave_tot[w];
        if (w < wave) wave_off += t;
        block_tot += t;
    }
    if (block_tot == 0) {                                     // workgroup-uniform: nothing visible here
        if (g < kp.P) kp.pair_off[vg] = 0;
        return;
    }
    // Returning atomics on ONE address are served one after the other (~0.2 us each): 117 workgroups per view on one
    // cursor cost this kernel 20 of its 42 us.  The arena is therefore cut into nseg segments with a cursor each;
    // workgroup b allocates from segment b % nseg (neighbouring workgroups hold mesh neighbours, so the fills stay even).
    const uint32_t seg = gb & (kp.nseg - 1u);
    if (tid == 0) s_base = seg * kp.seg_cap + atomicAdd(&kp.view_cursor[v * kCursorSegs + seg], block_tot);
    {   // bounding box (in tiles) of everything this workgroup touches
        int bx0 = tiles ? x0 : 0x7fffffff, by0 = tiles ? y0 : 0x7fffffff, bx1 = tiles ? x1 : 0, by1 = tiles ? y1 : 0;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            bx0 = min(bx0, __shfl_xor(bx0, d, 64)); by0 = min(by0, __shfl_xor(by0, d, 64));
            bx1 = max(bx1, __shfl_xor(bx1, d, 64)); by1 = max(by1, __shfl_xor(by1, d, 64));
        }
        if (lane == 0) { atomicMin(&s_bb[0], bx0); atomicMin(&s_bb[1], by0); atomicMax(&s_bb[2], bx1); atomicMax(&s_bb[3], by1); }
    }
    __syncthreads();
    const uint32_t pbase = s_base + wave_off + incl - tiles;
    // a Gaussian whose slots do not fit into its segment loses all of them (pair_off = cap fails every later bounds
    // test); k_scan_tiles raises the overflow flag from the cursors
    const bool fits = pbase + tiles <= (seg + 1u) * kp.seg_cap;
    if (g < kp.P) kp.pair_off[vg] = fits ? pbase : kp.cap;
    po.tiles = tiles; po.pbase = pbase; po.x0 = x0; po.y0 = y0; po.x1 = x1; po.y1 = y1; po.fits = fits;

    // ---- per-tile counts and the rank of every pair inside its tile ----
    // Gaussians of one workgroup are usually neighbours on the mesh, so they hit few distinct tiles: count them in an
    // LDS histogram over the workgroup's tile bounding box and send ONE returning global atomic per touched tile
    // (instead of one per pair).  Bounding boxes larger than the histogram fall back to per-pair global atomics.
    uint32_t *cnt = kp.tile_count + (size_t)v * kp.T;
    uint32_t *prank = kp.pair_rank + (size_t)v * kp.cap;
    const int bbx = s_bb[0], bby = s_bb[1], bw = s_bb[2] - s_bb[0], bh = s_bb[3] - s_bb[1];
    const int area = bw * bh;
    if (area <= kHist) {
        for (int i = tid; i < area; i += kBlock) s_hist[i] = 0;
        __syncthreads();
        for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++) atomicAdd(&s_hist[(y - bby) * bw + (x - bbx)], 1u);
        __syncthreads();
        for (int i = tid; i < area; i += kBlock) {
            const uint32_t c = s_hist[i];
            const int ty = i / bw, tx = i - ty * bw;
            s_hbase[i] = c ? atomicAdd(&cnt[(bby + ty) * kp.gx + bbx + tx], c) : 0u;
            s_hist[i] = 0;
        }
        __syncthreads();
        uint32_t pr = pbase;
        for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++, pr++) {
                const int i = (y - bby) * bw + (x - bbx);
                const uint32_t r = s_hbase[i] + atomicAdd(&s_hist[i], 1u);
                if (fits) prank[pr] = r;
            }
    } else {
        uint32_t pr = pbase;
        for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++, pr++) {
                const uint32_t r = atomicAdd(&cnt[y * kp.gx + x], 1u);
                if (fits) prank[pr] = r;
            }
    }
}

__global__ __launch_bounds__(kBlock) void k_preprocess(const KP kp)
{
    const uint32_t nblocks = (uint32_t)(kp.P + kBlock - 1) / kBlock;
    uint32_t gb, vb;
    vb = blockIdx.x / nblocks; gb = blockIdx.x - vb * nblocks;
    if (vb >= (uint32_t)kp.V) return;
    PreOut po;
    preprocess_body(kp, gb, vb, po);
}

// ---------------------------------------------------------------------------------------------------------
// A.2 per-view exclusive scan of tile counts
// ---------------------------------------------------------------------------------------------------------
// launch-order class of a tile: longest lists first, empty tiles last
__device__ __forceinline__ int count_bucket(const uint32_t c)
{
    if (c == 0) return kBuckets - 1;
    return (kBuckets - 2) - min(kBuckets - 2, 31 - __clz((int)c));
}

// Per-tile kernels walk the length-ordered tile list with a grid-stride loop (grid size: tile_grid() on the host; a
// fully resident grid was measured slower than the hardware dispatcher's dynamic balancing: tools/experiments/README.md).
// Heavy tiles start first and consecutive heavy tiles land on different XCDs (block b runs on XCD b % 8); empty tiles sit at
// the end of the list and end the loop.
struct TileOrder {
    uint32_t pre[kBuckets + 1];     // exclusive prefix of the bucket totals (wave-uniform, lives in SGPRs)
};

__device__ __forceinline__ void load_tile_order(const KP &kp, TileOrder &o)
{
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < kBuckets; k++) {
        o.pre[k] = acc;
        acc += kp.bucket_fill[k];
    }
    o.pre[kBuckets] = acc;
}

__device__ __forceinline__ uint32_t tile_order_id(const KP &kp, const TileOrder &o, const uint32_t b)
{
    int k = 0;
    uint32_t base = 0;
#pragma unroll
    for (int i = 1; i < kBuckets; i++)                               // static indices only: pre[] must stay in registers
        if (b >= o.pre[i]) { k = i; base = o.pre[i]; }
    return kp.order[(size_t)k * kp.V * kp.T + (b - base)];           // (view << 20) | tile
}

// Dense passes have tens of thousands of tiles per view (48,128 at 4096x3008): the scan is cut into chunks of kScanChunk tiles,
// one workgroup each.  k_tile_chunk_sums (launched only when there is more than one chunk) adds up every chunk; a chunk's
// workgroup of k_scan_tiles then starts from the sum of the chunks before it.
__global__ __launch_bounds__(kScanChunk) void k_tile_chunk_sums(const KP kp)
{
    __shared__ uint32_t s_w[kScanChunk / 64];
    const int v = blockIdx.y, c = blockIdx.x, tid = threadIdx.x, t = c * kScanChunk + tid;
    uint32_t x = t < kp.T ? kp.tile_count[(size_t)v * kp.T + t] : 0u;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) x += (uint32_t)__shfl_xor((int)x, d, 64);
    if ((tid & 63) == 0) s_w[tid >> 6] = x;
    __syncthreads();
    if (tid == 0) {
        uint32_t tot = 0;
#pragma unroll
        for (int w = 0; w < kScanChunk / 64; w++) tot += s_w[w];
        kp.chunk_sum[(size_t)v * kp.n_chunks + c] = tot;
    }
}

__global__ __launch_bounds__(kScanChunk) void k_scan_tiles(const KP kp)
{
    __shared__ uint32_t s_wave_tot[16], s_wave_max[16];
    __shared__ uint32_t s_carry;
    __shared__ uint32_t s_bcnt[kBuckets], s_bbase[kBuckets];
    const int v = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t *cnt = kp.tile_count + (size_t)v * kp.T;
    uint32_t *off = kp.tile_off + (size_t)v * kp.T;
    if (chunk == 0) {
        if (tid == 0) s_carry = 0;
    } else if (wave == 0) {                                 // pairs in the chunks before this one (64 chunks per round)
        uint32_t part = 0;
        for (int i = lane; i < chunk; i += 64) part += kp.chunk_sum[(size_t)v * kp.n_chunks + i];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) part += (uint32_t)__shfl_xor((int)part, d, 64);
        if (lane == 0) s_carry = part;
    }
    if (tid < kBuckets) s_bcnt[tid] = 0;
    __syncthreads();
    const int t = chunk * kScanChunk + tid;
    const uint32_t c = t < kp.T ? cnt[t] : 0u;
    const uint32_t incl = wave_incl_scan(c);
    uint32_t longest = c;                                   // longest list of the chunk -> status (policy input of the host)
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) longest = max(longest, (uint32_t)__shfl_xor((int)longest, d, 64));
    if (lane == 63) { s_wave_tot[wave] = incl; s_wave_max[wave] = longest; }
    // launch order: bucket the tiles of this chunk by list length
    const int bk = count_bucket(c);
    uint32_t r = 0;
    if (t < kp.T) r = atomicAdd(&s_bcnt[bk], 1u);
    __syncthreads();
    uint32_t woff = 0, tot = 0;
    longest = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) {
        const uint32_t x = s_wave_tot[w];
        if (w < wave) woff += x;
        tot += x;
        longest = max(longest, s_wave_max[w]);
    }
    const uint32_t carry = s_carry;
    if (t < kp.T) off[t] = carry + woff + incl - c;
    if (tid < kBuckets) s_bbase[tid] = s_bcnt[tid] ? atomicAdd(&kp.bucket_fill[tid], s_bcnt[tid]) : 0u;
    if (tid == 0 && longest > 0) atomicMax(&kp.status->max_tile_pairs, longest);
    __syncthreads();
    if (t < kp.T) kp.order[(size_t)bk * kp.V * kp.T + s_bbase[bk] + r] = ((uint32_t)v << 20) | (uint32_t)t;
    if (tid == 0 && chunk == kp.n_chunks - 1) {
        const uint32_t total = carry + tot;
        kp.view_total[v] = total;
        uint32_t fill = 0;                                  // fullest pair-slot segment of this view
        for (uint32_t k = 0; k < kp.nseg; k++) fill = max(fill, kp.view_cursor[v * kCursorSegs + k]);
        // capacity this view needs: every segment must hold the fullest one
        const unsigned long long need = max((unsigned long long)total, (unsigned long long)fill * kp.nseg);
        atomicMax(&kp.status->max_pairs, (uint32_t)min(need, 0xffffffffull));
        atomicAdd(&kp.status->total_pairs, (unsigned long long)total);
        if (total > kp.cap || fill > kp.seg_cap) atomicOr(&kp.status->overflow, 1u);
    }
}

// ---------------------------------------------------------------------------------------------------------
// A.2 scatter keys into tile bins
// ---------------------------------------------------------------------------------------------------------
// segmented backward (kSeg): one slot-table entry per kSeg list positions of a tile, at the slots the tile owns
// BIG: called from a launch that may be a big one-view launch, where only the tiles of at least kp.seg_min_pairs pairs are cut into
// segments (the one-view kernels of at most 1,024 tiles below never are: they do not even read the field)
template <bool BIG>
__device__ __forceinline__ void write_segment_slots(const KP &kp, const uint32_t id, const uint32_t off, const uint32_t n)
{
    if (kp.slots_per_view == 0u || n == 0u) return;
    if (BIG && n < kp.seg_min_pairs) return;
    const uint32_t nseg = (n + (1u << kp.seg_shift) - 1u) >> kp.seg_shift;
    const size_t first = (size_t)(id >> 20) * kp.slots_per_view + seg_slot0(kp, off, id & 0xfffffu);
    uint4 *tab = kp.slot_tab + first;
    for (uint32_t j = 0; j < nseg; j++) tab[j] = make_uint4(id, off, n, j | 0x80000000u);
    if (BIG && kp.slots_by_offset) {
        // the long tiles' segments, compact: the kernels that work on them take entry k, k + grid, ... of this list - even shares
        // and no walk over the (mostly empty) table.  One returning atomic per long tile (a few hundred per view).
        const uint32_t at = atomicAdd(&kp.status->live_segments, nseg);
        for (uint32_t j = 0; j < nseg; j++) kp.live[at + j] = (uint32_t)first + j;      // (one view: the index fits)
    }
}

__global__ __launch_bounds__(kBlock) void k_scatter(const KP kp)
{
    // One launch index: V * nb8 scatter workgroups (nb8 = blocks of 256 Gaussians, rounded up to eight), view after view in block
    // order, then the workgroups that flatten the tile list.  (Round 3 tried giving every XCD a contiguous eighth of each view's
    // blocks, so that the partial cache lines of a tile bin meet in one L2: config 4 went from 101 to 133 us.)
    const uint32_t nb8 = gaussian_grid(kp.P, 1);
    const uint32_t n_scatter = nb8 * (uint32_t)kp.V;
    if (blockIdx.x >= n_scatter) {
        // Tail blocks of this launch: flatten the length-ordered tile list into one 16-byte record per work item,
        // items[b] = (view << 20 | tile, arena offset, list length, pair count), so that a per-tile workgroup starts with ONE
        // scalar load instead of a chain of dependent loads (there are ~25k such workgroups per launch).
        const uint32_t b = (blockIdx.x - n_scatter) * kBlock + threadIdx.x;
        if (b >= (uint32_t)(kp.V * kp.T)) return;
        TileOrder ord;
        load_tile_order(kp, ord);
        const uint32_t id = tile_order_id(kp, ord, b);
        const size_t vt = (size_t)(id >> 20) * kp.T + (id & 0xfffffu);
        const uint32_t off = kp.tile_off[vt];
        const uint32_t n = off >= kp.cap ? 0u : min(kp.tile_count[vt], kp.cap - off);
        kp.items[b] = make_uint4(id, off, n, kp.tile_count[vt]);       // .w = 0: a truly empty tile (n = 0 also after an arena overflow)
        write_segment_slots<true>(kp, id, off, n);
        return;
    }
    const int v = (int)(blockIdx.x / nb8);
    const int g = (int)(blockIdx.x - (uint32_t)v * nb8) * kBlock + threadIdx.x;
    if (g >= kp.P) return;
    const size_t vg = (size_t)v * kp.P + g;
    const int r = kp.radii[vg];
    const float2 p = kp.xy[vg];                      // (requested with the radius, not behind it: one round trip less)
    const float dep = kp.depth[vg];
    uint32_t pr = kp.pair_off[vg];
    if (r <= 0) return;
    int x0, y0, x1, y1;
    tile_rect(p.x, p.y, r, kp.gx, kp.gy, x0, y0, x1, y1);
    const unsigned long long key = ((unsigned long long)__float_as_uint(dep) << 32) | (uint32_t)g;
    const uint32_t *off = kp.tile_off + (size_t)v * kp.T;
    const uint32_t *prank = kp.pair_rank + (size_t)v * kp.cap;
    unsigned long long *keys = kp.keys + (size_t)v * kp.cap;
    for (int y = y0; y < y1; y++)
        for (int x = x0; x < x1; x++, pr++) {
            if (pr >= kp.cap) return;
            const uint32_t pos = off[y * kp.gx + x] + prank[pr];     // rank inside the tile was fixed by k_preprocess
            if (pos < kp.cap) keys[pos] = key;
        }
}

// ---------------------------------------------------------------------------------------------------------
// A.2 for ONE view of at most 1,024 tiles (the reference's own call shape: train.py:661-673 renders one 512x375 view = 768
// tiles per iteration): scan and scatter in one launch.  A launch that small is made of kernel boundaries, not of work -
// k_scan_tiles lasts 6.7 us there for 768 additions - and nothing in the scan needs another workgroup: every scatter
// workgroup adds up the view's tile counts itself (four per thread, in LDS) and takes its offsets from there; one extra
// workgroup does what else the scan kernel leaves behind - the offsets in memory, the view's total, the status block and
// the length-ordered work items (built in LDS: a single workgroup sees every tile, so the per-class lists of k_scan_tiles
// and the flattening pass of k_scatter are not needed).
// ---------------------------------------------------------------------------------------------------------
constexpr int kSmallTiles = 4 * kBlock;

// T4D_FLAG_ASYNC_STATUS on a one-view launch: the 16-byte status { overflow, max pairs per view, total pairs } goes to the caller's
// PINNED host memory straight from the thread that knows it - two system-scope stores instead of a copy kernel on the stream
// (3-5 us of GPU time and a launch per forward of Topo4D's loop).  The host treats a block as landed when neither word holds its
// sentinel, so the order of the two stores does not matter.
__device__ __forceinline__ void publish_status(const KP &kp, const uint32_t overflow, const uint32_t max_pairs, const unsigned long long total)
{
    if (kp.host_status == nullptr) return;
    __hip_atomic_store(&kp.host_status[1], total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&kp.host_status[0], (unsigned long long)overflow | ((unsigned long long)max_pairs << 32), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(kBlock) void k_scan_scatter_small(const KP kp)
{
    __shared__ uint32_t s_off[kSmallTiles];
    __shared__ uint32_t s_wtot[4];
    __shared__ uint32_t s_bcnt[kBuckets], s_bpre[kBuckets];
    __shared__ uint32_t s_longest[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // ---- exclusive scan of the tile counts: thread t owns tiles 4t .. 4t + 3
    uint32_t c[4];
#pragma unroll
    for (int j = 0; j < 4; j++) c[j] = 4 * tid + j < kp.T ? kp.tile_count[4 * tid + j] : 0u;
    const uint32_t mine = (c[0] + c[1]) + (c[2] + c[3]);
    const uint32_t incl = wave_incl_scan(mine);
    if (lane == 63) s_wtot[wave] = incl;
    if (tid < kBuckets) s_bcnt[tid] = 0;
    __syncthreads();
    uint32_t base = incl - mine, total = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const uint32_t x = s_wtot[w];
        if (w < wave) base += x;
        total += x;
    }
    uint32_t off[4];
    off[0] = base; off[1] = off[0] + c[0]; off[2] = off[1] + c[1]; off[3] = off[2] + c[2];
#pragma unroll
    for (int j = 0; j < 4; j++) s_off[4 * tid + j] = off[j];
    const uint32_t nb8 = gaussian_grid(kp.P, 1);
    if (blockIdx.x == nb8) {
        // ---- the scan kernel's other products, and the work items
        uint32_t longest = max(max(c[0], c[1]), max(c[2], c[3]));
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) longest = max(longest, (uint32_t)__shfl_xor((int)longest, d, 64));
        if (lane == 0) s_longest[wave] = longest;
        int bk[4];
        uint32_t rank[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            bk[j] = count_bucket(c[j]);
            rank[j] = 4 * tid + j < kp.T ? atomicAdd(&s_bcnt[bk[j]], 1u) : 0u;
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t acc = 0;
            for (int k = 0; k < kBuckets; k++) { s_bpre[k] = acc; acc += s_bcnt[k]; }
        }
        __syncthreads();
        if (tid < kBuckets) kp.bucket_fill[tid] = s_bcnt[tid];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int t = 4 * tid + j;
            if (t < kp.T) {
                kp.tile_off[t] = off[j];
                const uint32_t n = off[j] >= kp.cap ? 0u : min(c[j], kp.cap - off[j]);
                kp.items[s_bpre[bk[j]] + rank[j]] = make_uint4((uint32_t)t, off[j], n, c[j]);       // view 0: id = tile
                write_segment_slots<false>(kp, (uint32_t)t, off[j], n);
            }
        }
        if (tid == 0) {
            kp.view_total[0] = total;
            uint32_t fill = 0;                                  // fullest pair-slot segment (see k_scan_tiles)
            for (uint32_t k = 0; k < kp.nseg; k++) fill = max(fill, kp.view_cursor[k]);
            const unsigned long long need = max((unsigned long long)total, (unsigned long long)fill * kp.nseg);
            kp.status->max_pairs = (uint32_t)min(need, 0xffffffffull);
            kp.status->total_pairs = (unsigned long long)total;
            kp.status->max_tile_pairs = max(max(s_longest[0], s_longest[1]), max(s_longest[2], s_longest[3]));
            const uint32_t ovf = (total > kp.cap || fill > kp.seg_cap) ? 1u : 0u;
            if (ovf) kp.status->overflow = 1u;
            publish_status(kp, ovf, (uint32_t)min(need, 0xffffffffull), (unsigned long long)total);
        }
        return;
    }
    __syncthreads();
    // ---- scatter (k_scatter's body on the offsets in LDS)
    const int g = (int)blockIdx.x * kBlock + tid;
    if (g >= kp.P) return;
    const int r = kp.radii[g];
    const float2 p = kp.xy[g];
    const float dep = kp.depth[g];
    uint32_t pr = kp.pair_off[g];
    if (r <= 0) return;
    int x0, y0, x1, y1;
    tile_rect(p.x, p.y, r, kp.gx, kp.gy, x0, y0, x1, y1);
    const unsigned long long key = ((unsigned long long)__float_as_uint(dep) << 32) | (uint32_t)g;
    for (int y = y0; y < y1; y++)
        for (int x = x0; x < x1; x++, pr++) {
            if (pr >= kp.cap) return;
            const uint32_t pos = s_off[y * kp.gx + x] + kp.pair_rank[pr];
            if (pos < kp.cap) kp.keys[pos] = key;
        }
}

// ---------------------------------------------------------------------------------------------------------
// The whole binning front end of ONE small view (Topo4D's own call shape) in one launch: preprocess, then - behind one grid-wide
// barrier - what k_scan_scatter_small does, on the values the threads still hold (tile rectangle, depth, pair slots).  The
// launch is at most 128 workgroups of 256 threads: all of them are resident at once, so a spin barrier is safe.  What crosses
// the barrier between workgroups are the per-tile counts and the slot cursors, both products of RETURNING device-scope atomics
// (performed at the memory side, complete before their result is used) and read back with agent-scope atomic loads: no fence,
// no L2 write-back (a __threadfence() per workgroup cost the round-3 experiment 10x its gain).  One launch and one trip
// through memory less per forward: 9.9 + 7.7 us -> see HISTORY.md section 5.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t load_agent(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(kBlock) void k_front_small(const KP kp)
{
    __shared__ uint32_t s_off[kSmallTiles];
    __shared__ uint32_t s_wtot[4];
    __shared__ uint32_t s_bcnt[kBuckets], s_bpre[kBuckets];
    __shared__ uint32_t s_longest[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t nb8 = gaussian_grid(kp.P, 1), nblocks = (uint32_t)(kp.P + kBlock - 1) / kBlock;
    PreOut po;
    po.tiles = 0; po.pbase = 0; po.x0 = po.y0 = po.x1 = po.y1 = 0; po.depth = 0.f; po.fits = false;
    if (blockIdx.x < nblocks) preprocess_body(kp, blockIdx.x, 0u, po);
    // ---- the grid-wide barrier: every count of this workgroup has been added (the atomics returned) when thread 0 arrives
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_fetch_add(&kp.status->grid_sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (load_agent(&kp.status->grid_sync) < nb8 + 1u) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
    // ---- exclusive scan of the tile counts: thread t owns tiles 4t .. 4t + 3 (as k_scan_scatter_small)
    uint32_t c[4];
#pragma unroll
    for (int j = 0; j < 4; j++) c[j] = 4 * tid + j < kp.T ? load_agent(&kp.tile_count[4 * tid + j]) : 0u;
    const uint32_t mine = (c[0] + c[1]) + (c[2] + c[3]);
    const uint32_t incl = wave_incl_scan(mine);
    if (lane == 63) s_wtot[wave] = incl;
    if (tid < kBuckets) s_bcnt[tid] = 0;
    __syncthreads();
    uint32_t base = incl - mine, total = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const uint32_t x = s_wtot[w];
        if (w < wave) base += x;
        total += x;
    }
    uint32_t off[4];
    off[0] = base; off[1] = off[0] + c[0]; off[2] = off[1] + c[1]; off[3] = off[2] + c[2];
#pragma unroll
    for (int j = 0; j < 4; j++) s_off[4 * tid + j] = off[j];
    if (blockIdx.x == nb8) {
        // ---- the scan kernel's other products, and the work items
        uint32_t longest = max(max(c[0], c[1]), max(c[2], c[3]));
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) longest = max(longest, (uint32_t)__shfl_xor((int)longest, d, 64));
        if (lane == 0) s_longest[wave] = longest;
        int bk[4];
        uint32_t rank[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            bk[j] = count_bucket(c[j]);
            rank[j] = 4 * tid + j < kp.T ? atomicAdd(&s_bcnt[bk[j]], 1u) : 0u;
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t acc = 0;
            for (int k = 0; k < kBuckets; k++) { s_bpre[k] = acc; acc += s_bcnt[k]; }
        }
        __syncthreads();
        if (tid < kBuckets) kp.bucket_fill[tid] = s_bcnt[tid];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int t = 4 * tid + j;
            if (t < kp.T) {
                kp.tile_off[t] = off[j];
                const uint32_t n = off[j] >= kp.cap ? 0u : min(c[j], kp.cap - off[j]);
                kp.items[s_bpre[bk[j]] + rank[j]] = make_uint4((uint32_t)t, off[j], n, c[j]);       // view 0: id = tile
                write_segment_slots<false>(kp, (uint32_t)t, off[j], n);
            }
        }
        if (tid == 0) {
            kp.view_total[0] = total;
            uint32_t fill = 0;                                  // fullest pair-slot segment (see k_scan_tiles)
            for (uint32_t k = 0; k < kp.nseg; k++) fill = max(fill, load_agent(&kp.view_cursor[k]));
            const unsigned long long need = max((unsigned long long)total, (unsigned long long)fill * kp.nseg);
            kp.status->max_pairs = (uint32_t)min(need, 0xffffffffull);
            kp.status->total_pairs = (unsigned long long)total;
            kp.status->max_tile_pairs = max(max(s_longest[0], s_longest[1]), max(s_longest[2], s_longest[3]));
            const uint32_t ovf = (total > kp.cap || fill > kp.seg_cap) ? 1u : 0u;
            if (ovf) kp.status->overflow = 1u;
            publish_status(kp, ovf, (uint32_t)min(need, 0xffffffffull), (unsigned long long)total);
        }
        return;
    }
    __syncthreads();
    // ---- scatter, from the registers of the preprocess pass (a Gaussian that lost its slots - arena overflow - scatters nothing)
    if (po.tiles == 0u || !po.fits) return;
    const int g = (int)blockIdx.x * kBlock + tid;
    const unsigned long long key = ((unsigned long long)__float_as_uint(po.depth) << 32) | (uint32_t)g;
    uint32_t pr = po.pbase;
    for (int y = po.y0; y < po.y1; y++)
        for (int x = po.x0; x < po.x1; x++, pr++) {
            if (pr >= kp.cap) return;
            const uint32_t pos = s_off[y * kp.gx + x] + kp.pair_rank[pr];
            if (pos < kp.cap) kp.keys[pos] = key;
        }
}

