// t4d_raster_render_bwd.h - part of the translation unit t4d_raster.hip (included there, inside its anonymous namespace; not a
// stand-alone header).  A.4: the ten-sum DPP transpose-reduce and the backward render kernel (whole tiles or depth segments).
// See t4d_raster.hip for the overview, the constants, the state layout and the kernel parameter block.
// ---------------------------------------------------------------------------------------------------------
// Reduction of TEN values over each 16-lane DPP row ("transpose-reduce"): at every butterfly level two partial-sum
// vectors are folded into one, each half of the lanes keeping a different value, so the work halves per level
// instead of staying at 10 adds x 4 levels.  Levels: xor8 by row_ror:8, xor4 by two bank-masked row shifts, xor2 /
// xor1 by quad_perm.  The four rows of a wave reduce four different splats at the same time.
// ---------------------------------------------------------------------------------------------------------
// In-place butterfly over ten VGPRs, written as one asm block: bank-masked DPP adds do the "keep one half, send the
// other" selection of the transpose for free (v_cndmask + v_mov_dpp pairs otherwise), and the instruction order keeps
// every DPP read at least two instructions behind the write of its source (the gfx9 VALU->DPP hazard), so no s_nop is
// needed inside; the leading s_nop covers inputs produced just before the block.
//   level xor8 (row_ror:8):  r[2m] <- r[2m + b3] summed over the pair          (banks 2,3 = lanes with b3 set)
//   level xor4 (row_shl/shr:4): r1 <- c_{b2}, r3 <- c_{2+b2}, r5 <- c_4        (banks 0,2 read lane+4; banks 1,3 lane-4)
//   levels xor2, xor1 (quad_perm): no bank masks at this granularity (a bank is four consecutive lanes), so the transposing
//   is done with selects on the constant lane masks b1 / b0: xor2 folds (r1, r3) into one register and r5 into itself, xor1
//   folds those two into ONE - seven instructions, and the caller needs no selection (plain butterflies on the three
//   registers plus the caller's two selects were eight).
// Returns, in lane i = (b3 b2 b1 b0) of a row: the sum of value  2*b2 + b3  (b1 b0 = 00),  4 + 2*b2 + b3  (b1 b0 = 10),
// 8 + b3  (b0 = 1; four lanes per half row hold it, row10_index picks b2 = b1 = 0).
// Operands: values 2 and 5 are read-only inputs whose sums go to fresh registers (o2, o5): r[1], r[2] (and r[3], r[5]) are the
// halves of ONE packed-multiply result, and tying both halves of a register pair to in/out operands costs a v_mov each.
template <bool NINE>          // NINE: r[9] is known to be zero (no depth cotangent): its banked add is skipped
__device__ __forceinline__ float reduce10_row(float (&r)[10])
{
    float o2, o5, ta, tb;
#define T4D_RED_HEAD                                                                                  \
        "s_nop 1\n\t"                                                                                 \
        "v_add_f32_dpp %[r0], %[r0], %[r0] row_ror:8 row_mask:0xf bank_mask:0xf\n\t"                  \
        "v_add_f32_dpp %[o2], %[r2], %[r2] row_ror:8 row_mask:0xf bank_mask:0xf\n\t"                  \
        "v_add_f32_dpp %[r4], %[r4], %[r4] row_ror:8 row_mask:0xf bank_mask:0xf\n\t"                  \
        "v_add_f32_dpp %[r6], %[r6], %[r6] row_ror:8 row_mask:0xf bank_mask:0xf\n\t"                  \
        "v_add_f32_dpp %[r8], %[r8], %[r8] row_ror:8 row_mask:0xf bank_mask:0xf\n\t"                  \
        "v_add_f32_dpp %[r0], %[r1], %[r1] row_ror:8 row_mask:0xf bank_mask:0xc\n\t"                  \
        "v_add_f32_dpp %[o2], %[r3], %[r3] row_ror:8 row_mask:0xf bank_mask:0xc\n\t"                  \
        "v_add_f32_dpp %[r4], %[r5], %[r5] row_ror:8 row_mask:0xf bank_mask:0xc\n\t"                  \
        "v_add_f32_dpp %[r6], %[r7], %[r7] row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
#define T4D_RED_TAIL                                                                                  \
        "v_add_f32_dpp %[r1], %[r0], %[r0] row_shl:4 row_mask:0xf bank_mask:0x5\n\t"                  \
        "v_add_f32_dpp %[r3], %[r4], %[r4] row_shl:4 row_mask:0xf bank_mask:0x5\n\t"                  \
        "v_add_f32_dpp %[o5], %[r8], %[r8] row_shl:4 row_mask:0xf bank_mask:0x5\n\t"                  \
        "v_add_f32_dpp %[r1], %[o2], %[o2] row_shr:4 row_mask:0xf bank_mask:0xa\n\t"                  \
        "v_add_f32_dpp %[r3], %[r6], %[r6] row_shr:4 row_mask:0xf bank_mask:0xa\n\t"                  \
        "v_add_f32_dpp %[o5], %[r8], %[r8] row_shr:4 row_mask:0xf bank_mask:0xa\n\t"                  \
        "v_cndmask_b32_e64 %[tb], %[r3], %[r1], %[m1]\n\t"              /* b1 ? r1 : r3  (goes to the partner) */ \
        "v_cndmask_b32_e64 %[ta], %[r1], %[r3], %[m1]\n\t"              /* b1 ? r3 : r1  (stays)               */ \
        "v_add_f32_dpp %[o5], %[o5], %[o5] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"        \
        "v_add_f32_dpp %[ta], %[tb], %[ta] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"        \
        "v_cndmask_b32_e64 %[tb], %[o5], %[ta], %[m0]\n\t"              /* b0 ? x : y  (goes to the partner)   */ \
        "v_cndmask_b32_e64 %[ta], %[ta], %[o5], %[m0]\n\t"              /* b0 ? y : x  (stays)                 */ \
        "s_nop 0\n\t"                                                                                 \
        "v_add_f32_dpp %[ta], %[tb], %[ta] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
#define T4D_RED_OUT [r0] "+v"(r[0]), [r1] "+v"(r[1]), [o2] "=&v"(o2), [r3] "+v"(r[3]), [r4] "+v"(r[4]), [o5] "=&v"(o5), \
                    [r6] "+v"(r[6]), [r8] "+v"(r[8]), [ta] "=&v"(ta), [tb] "=&v"(tb)
#define T4D_RED_MASKS [m1] "s"(0xccccccccccccccccull), [m0] "s"(0xaaaaaaaaaaaaaaaaull)
    if (NINE) {
        // r8 then holds the xor8 sum of value 8 in BOTH halves of the row; only the b3 = 0 lane is used (row10_index)
        asm(T4D_RED_HEAD T4D_RED_TAIL : T4D_RED_OUT : [r2] "v"(r[2]), [r5] "v"(r[5]), [r7] "v"(r[7]), T4D_RED_MASKS);
    } else {
        asm(T4D_RED_HEAD
            "v_add_f32_dpp %[r8], %[r9], %[r9] row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            T4D_RED_TAIL
            : T4D_RED_OUT : [r2] "v"(r[2]), [r5] "v"(r[5]), [r7] "v"(r[7]), [r9] "v"(r[9]), T4D_RED_MASKS);
    }
    return ta;
#undef T4D_RED_HEAD
#undef T4D_RED_TAIL
#undef T4D_RED_OUT
#undef T4D_RED_MASKS
}

// which of the ten sums lane i of a row holds after reduce10_row; -1 = none (or a duplicate)
__device__ __forceinline__ int row10_index(const int lane)
{
    const int b0 = lane & 1, b1 = (lane >> 1) & 1, b2 = (lane >> 2) & 1, b3 = (lane >> 3) & 1;
    if (!b0 && !b1) return 2 * b2 + b3;
    if (!b0 && b1) return 4 + 2 * b2 + b3;
    if (b0 && !b1 && !b2) return 8 + b3;
    return -1;
}

__device__ __forceinline__ uint32_t row_max_u32(uint32_t v)      // every lane gets the maximum over its 16-lane row
{
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true));   // row_half_mirror
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true));   // row_mirror
    return v;
}

// ---------------------------------------------------------------------------------------------------------
// A.4 backward replay.  No global atomics: one kGP-float record per (Gaussian,tile) pair.
// record (raw sums over the tile's pixels, e = G * dL/dalpha, d = splat centre - pixel):
//   [0] sum e   [1,2] sum e*d   [3,4,5] sum e*dx*dx, e*dx*dy, e*dy*dy   [6,7,8] sum alpha*T*dL/dC   [9] sum alpha*T*dL/dD
// Inside the workgroup every wave owns an LDS slab of ten sums per staged splat; a row's reduced sums are added to it
// by plain read-add-write (no LDS float atomics: they retire ~3 cycles per lane here), rows that hold the same splat
// in the same step taking turns, and the slabs of the four waves are summed in wave order when the batch is written
// out.  Every addition order is fixed, so the gradients are bit-reproducible.
// ---------------------------------------------------------------------------------------------------------
// DA = the caller supplied dL/ddepth and/or dL/dalpha.  Topo4D discards depth and alpha (train.py:307), so its backward
// runs the DA = false instantiation, which carries neither the two extra suffix accumulators nor their products.
#ifndef T4D_BWD_WAVES
#define T4D_BWD_WAVES 5                  // = workgroups per CU (30.8 KB of LDS each); 4 is 14 % slower, 6 spills (round-3 sweep)
#endif
#ifndef T4D_SEG_WAVES
#define T4D_SEG_WAVES 5          // (4 = 128 registers, no spills: config-2 scene 1 view 36.3 us, 3 views 64.4; 5: 37.4 / 59.3, 6 views 104.7 -> 95.6)
#endif
#ifndef T4D_BWD_DA_WAVES
#define T4D_BWD_DA_WAVES T4D_BWD_WAVES
#endif
#define T4D_BWD_NW (LAT ? 2 : (SEGN != 0 ? T4D_SEG_WAVES : (DA ? T4D_BWD_DA_WAVES : T4D_BWD_WAVES)))
#define T4D_BWD_ATTR __attribute__((amdgpu_waves_per_eu(LAT ? 1 : T4D_BWD_NW, T4D_BWD_NW)))
constexpr int kAcc = 10;                 // sums per (wave, staged splat) slab entry
constexpr int kEmptySpan = 64;           // tiles per spare workgroup of the empty-tile share of cotangent_dot
// LAT: the latency build (see k_render_fwd): one slab per DPP ROW instead of one per wave (82 KB of LDS: one workgroup per CU
// is all such a launch has anyway), so two rows holding the same splat in the same step never meet and the conflict
// detection and its branches disappear; the gradient arithmetic is predicated with selects instead of an exec-masked region,
// which lets the compiler interleave the four steps of a group.
// SEG: the segmented backward of small launches (kSeg): a work item is ONE segment of a tile list - workgroup b takes slot b of
// the slot table - and the replay starts from the forward's snapshot at the segment's far end instead of from the list's end.
// LONG: the two launches of a big one-view launch (seg_mode 2) - the whole-tile build (SEGN == 0) leaves out the tiles that own
// segments, the segmented build (SEGN != 0) walks the mostly empty slot table with a fixed number of workgroups.  A template
// parameter so that every other launch shape runs the code it always ran (a one-view launch of Topo4D's size lasts 23 us: the same
// tests at run time in its prologue cost 0.8-1.6 us).
template <bool DA, bool LAT, int SEGN, bool LONG = false>
__global__ __launch_bounds__(kBlock) T4D_BWD_ATTR void k_render_bwd(const KP kp)
{
    constexpr bool SEG = SEGN != 0;
    // a segment is ONE staged batch: the segmented build stages SEGN splats per round (128, or 64 for a one-view launch), the
    // whole-tile builds kBwdBatch (these shadow the globals inside the kernel)
    constexpr int kSeg = SEG ? SEGN : ::kSeg;
    constexpr int kBwdBatch = SEG ? SEGN : ::kBwdBatch;
    constexpr int kSlabs = LAT ? 16 : 4;
    constexpr int kChunks = (kBwdBatch + 63) / 64;
    constexpr int kListStride = kBwdBatch + 4;
    // A staged splat is ONE 40-byte record - scaled conic + opacity (16) | rgb + depth (16) | xy (8) - exactly as long as a slab
    // entry (ten floats), and list entries are slot * 40: the byte offset of BOTH, so a step spends no vector instruction on
    // addresses (records are read as 8-byte words: a 40-byte stride keeps them 8- but not 16-byte aligned).
    constexpr int kEnt = 40;
    static_assert(kBwdBatch % 64 == 0, "staged slots come in chunks of one per lane");
    static_assert(kGP == kAcc, "the slab entry and the scratch record hold the same ten sums");
    static_assert(kAcc * 4 == kEnt, "a slab entry and a staged record must have the same stride");
    // One struct, so that the layout is ours: the staged records sit at LDS offset 0 and the replay's paired 8-byte reads reach
    // them with immediate offsets (behind the slabs, at 20 KiB, every step paid a vector add for the address).
    struct __attribute__((aligned(16))) Shared {
        unsigned char rec[(kBwdBatch + 1) * kEnt];
        float acc[kSlabs][kBwdBatch + 1][kAcc];                 // + the null splat's (never read) row
        unsigned short list[4][4][kListStride];
        float cut_r2[kChunks * 64];                             // cut-off of every staged splat (< 0: none in this slot)
        uint32_t pair[kBwdBatch];
        uint32_t wmax[4];
    };
    static_assert(((kBwdBatch + 1) * kEnt) % 8 == 0 && (sizeof(float) * kSlabs * (kBwdBatch + 1) * kAcc) % 8 == 0 &&
                  (sizeof(unsigned short) * 16 * kListStride) % 8 == 0, "8-byte members must stay 8-byte aligned");
    __shared__ Shared sh;
    auto &s_rec = sh.rec;
    auto &s_pair = sh.pair;
    auto &s_acc = sh.acc;
    auto &s_r2 = sh.cut_r2;
    auto &s_wmax = sh.wmax;
    auto &s_list = sh.list;
    constexpr int kNull = kBwdBatch;

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, row = lane >> 4;
    // without a depth cotangent the ninth pair is not transposed (reduce10_row<true>): the lane that would hold sum 9 holds a
    // second copy of sum 8 and must stay out
    const int my_slot = (!DA && row10_index(lane & 15) == 9) ? -1 : row10_index(lane & 15);
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < kEnt / 8; k++) reinterpret_cast<float2 *>(s_rec + kNull * kEnt)[k] = make_float2(0.f, 0.f);
    }
    if (blockIdx.x >= kp.tile_blocks) {
        // Spare workgroups behind the tile workgroups, launched only when the caller asked for <outputs, cotangents>: the EMPTY
        // tiles' share.  An empty tile shows the background at T = 1, so on black (Topo4D: helpers.py setup_camera, bg = 0)
        // there is nothing to add and the workgroup leaves at once; otherwise it sums bg . dL/dC over its kEmptySpan tiles.
        const uint32_t spans = (uint32_t)(kp.T + kEmptySpan - 1) / kEmptySpan;
        const uint32_t j = blockIdx.x - kp.tile_blocks;
        const int v = (int)(j / spans), t0 = (int)(j % spans) * kEmptySpan;
        const float *vb = kp.views + (size_t)v * T4D_VIEW_FLOATS + 35;
        const float b0 = vb[0], b1 = vb[1], b2 = vb[2];
        if (b0 == 0.f && b1 == 0.f && b2 == 0.f) return;
        const size_t HWe = (size_t)kp.H * kp.W;
        const float *dc = kp.dL_dcolor + (size_t)v * 3 * HWe;
        for (int t = t0; t < min(t0 + kEmptySpan, kp.T); t++) {
            if (kp.tile_count[(size_t)v * kp.T + t] != 0u) continue;           // workgroup-uniform
            const int ty = t / kp.gx, tx = t - ty * kp.gx;
            int ex, ey;
            tile_pixel(tid, tx, ty, ex, ey);
            float d = 0.f;
            if (ex < kp.W && ey < kp.H) {
                const size_t pe = (size_t)ey * kp.W + ex;
                d = fmaf(b0, dc[pe], fmaf(b1, dc[HWe + pe], b2 * dc[2 * HWe + pe]));
            }
            d = wave_sum_to_lane63(d);
            if (lane == 63) kp.tile_dot[((size_t)v * kp.T + t) * 4 + wave] = d;
        }
        return;
    }
    uint4 it;
    if (SEG && !LONG) {
        it = kp.slot_tab[blockIdx.x];                // one slot per workgroup; most slots hold no segment
        if (it.w == 0u) return;
    }
    for (int i = tid; i < kSlabs * (kBwdBatch + 1) * kAcc; i += kBlock) (&s_acc[0][0][0])[i] = 0.f;   // slabs are all-zero between batches
    // work items: the length-ordered tile list (whole tiles), ONE slot of the slot table (segments of a small launch), or a strided
    // walk over the mostly empty table of a big one-view launch (LONG)
    // (LONG segments: entries k, k + grid, ... of the compact list of live segments - the table itself is mostly empty, and a
    // strided walk over it, even with one gather per 64 slots, cost 20-40 us and dealt the segments out unevenly)
    for (uint32_t item = blockIdx.x; item < (SEG ? (LONG ? kp.status->live_segments : blockIdx.x + 1u) : (uint32_t)(kp.V * kp.T)); item += kp.tile_blocks) {
    if (SEG && LONG) it = kp.slot_tab[kp.live[item]];
    if (!SEG) it = kp.items[item];
    const int seg_j = SEG ? (int)(it.w & 0x7fffffffu) : 0;           // this item's segment: list positions [seg_j kSeg, (seg_j + 1) kSeg)
    const int v = (int)(it.x >> 20), t_ = (int)(it.x & 0xfffffu);
    const int ty = t_ / kp.gx, tx = t_ - ty * kp.gx;
    const uint32_t off = it.y, n = it.z;
    if (n == 0) break;                                             // ordered by length: only empty tiles remain
    // a big one-view launch: the tiles the forward cut into segments (it wrote their slot-table entries with their snapshots) are
    // the segmented launch's
    if (!SEG && LONG && n >= kp.seg_min_pairs &&
        kp.slot_tab[(size_t)v * kp.slots_per_view + seg_slot0(kp, off, (uint32_t)t_)].w != 0u) continue;
    const unsigned long long *keys = kp.keys + (size_t)v * kp.cap + off;
    const float *r2_in = kp.cut_r2 + (size_t)v * kp.cap + off;
    const float2 *xy = kp.xy + (size_t)v * kp.P;
    const float4 *co = kp.conic_opacity + (size_t)v * kp.P;
    const float *rgb = kp.shs ? kp.rgb + (size_t)v * kp.P * 3 : kp.colors_precomp + 3 * param_row0(kp, v);
    const int32_t *radii = kp.radii + (size_t)v * kp.P;
    const uint32_t *pair_off = kp.pair_off + (size_t)v * kp.P;
    float2 *grad_pair = reinterpret_cast<float2 *>(kp.grad_pair) + (size_t)v * kp.cap * (kGP / 2);
    const float *vr = kp.views + (size_t)v * T4D_VIEW_FLOATS;

    int px, py;
    tile_pixel(tid, tx, ty, px, py);
    const bool inside = px < kp.W && py < kp.H;
    const v2f pix_f = { (float)px, (float)py };
    const size_t HW = (size_t)kp.H * kp.W, pix = (size_t)py * kp.W + px;

    float T_final = 0.f, dp0 = 0.f, dp1 = 0.f, dp2 = 0.f, ddep = 0.f, dalp = 0.f;
    uint32_t last_contributor = 0;
    if (inside) {
        T_final = kp.final_T[(size_t)v * HW + pix];
        last_contributor = kp.n_contrib[(size_t)v * HW + pix];
        const float *dc = kp.dL_dcolor + (size_t)v * 3 * HW;
        dp0 = dc[pix]; dp1 = dc[HW + pix]; dp2 = dc[2 * HW + pix];
        if (DA && kp.dL_ddepth) ddep = kp.dL_ddepth[(size_t)v * HW + pix];
        if (DA && kp.dL_dalpha) dalp = kp.dL_dalpha[(size_t)v * HW + pix];
    }
    const v2f dp01 = { dp0, dp1 };
    float T = T_final;
    // Suffix state of the replay.  Upstream keeps one running "colour behind me" per channel (+ depth, + alpha) and dots
    // it with dL/dpixel afterwards; the recursion is linear, so the dot product is taken FIRST and a single scalar is
    // carried:  q_i = c_i . dL/dC (+ depth_i dL/dD + dL/dAlpha),  acc <- alpha_i q_i + (1 - alpha_i) acc  once splat i is done
    // (upstream applies the same update lazily, at the next contributor).
    // The BACKGROUND is the splat behind all others (colour bg, alpha 1): the recursion starts from its q = bg . dL/dC instead of
    // from zero.  Upstream starts from zero and subtracts T_final / (1 - alpha_i) * (bg . dL/dC) from every dL/dalpha_i; with
    // acc' = acc + T_final (bg . dL/dC) / T_i (T_i = transmittance in front of splat i) both the update acc' <- alpha q + (1 - alpha) acc'
    // and dL/dalpha_i = (q_i - acc') T_i hold exactly - one multiply and one fused multiply-add less per step, and for a black
    // background (Topo4D: helpers.py setup_camera, bg = 0) the same bits as before.
    float acc = vr[35] * dp0 + vr[36] * dp1 + vr[37] * dp2;
    const int nb = (int)((n + kBwdBatch - 1) / kBwdBatch);
    if (SEG && seg_j + 1 < nb) {
        // A segment that does not end at the list's end starts from the forward's snapshot at position p = (seg_j + 1) kSeg:
        // T = the transmittance in front of p, acc = the colour behind p as the recursion would hold it there,
        // ((C_final - C_prefix(p)) . dL/dC (+ depth and alpha terms) + T_final bg . dL/dC) / T(p).  A pixel whose last contributor
        // lies before p has its final state at p: exactly the start values above (the forward writes no snapshot for a
        // finished wave, so nothing is read for such a pixel).
        const uint32_t p = (uint32_t)(seg_j + 1) * kSeg;
        if (last_contributor > p) {
            const float *sb = kp.snap + ((size_t)v * kp.slots_per_view + seg_slot0(kp, off, (uint32_t)t_)) * (kSnapFloats * kBlock) + tid;
            const float *sp = sb + (size_t)seg_j * (kSnapFloats * kBlock), *sf = sb + (size_t)(nb - 1) * (kSnapFloats * kBlock);
            const float Tp = sp[0];
            float suf = fmaf(sf[kBlock] - sp[kBlock], dp0, fmaf(sf[2 * kBlock] - sp[2 * kBlock], dp1, (sf[3 * kBlock] - sp[3 * kBlock]) * dp2));
            if (DA) suf = fmaf(sf[4 * kBlock] - sp[4 * kBlock], ddep, suf) + (Tp - T_final) * dalp;
            acc = fmaf(T_final, acc, suf) / Tp;
            T = Tp;
        }
    }

    const uint32_t rmax_v = row_max_u32(last_contributor);
    uint32_t row_max[4];
#pragma unroll
    for (int r = 0; r < 4; r++) row_max[r] = lane_value(rmax_v, 16 * r);
    const uint32_t wave_max = max(max(row_max[0], row_max[1]), max(row_max[2], row_max[3]));
    if (lane == 0) s_wmax[wave] = wave_max;
    __syncthreads();
    const uint32_t tile_max = max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3]));

    for (int bi = SEG ? seg_j : nb - 1; bi >= (SEG ? seg_j : 0); bi--) {
        const uint32_t lo = (uint32_t)bi * kBwdBatch;
        const int cnt = (int)min((uint32_t)kBwdBatch, n - lo);
        const bool live = lo < tile_max;      // workgroup-uniform
        // ---- stage ----
        if (tid < cnt) s_pair[tid] = 0xffffffffu;
        if (tid < kChunks * 64) {
            float r2 = -1.f;                                 // a slot without a splat touches nothing ...
            float2 p = make_float2(0.f, 0.f);                // ... and holds a finite centre
            if (tid < cnt) {
                // ONE level of dependent loads behind the key: everything a splat needs is requested before any of it is used
                // (as the code was written - centre and radius, then the pair slot, then conic and colour - the staging waves went
                // through four dependent round trips per batch while the other waves waited at the barrier)
                const unsigned long long key = keys[lo + tid];
                const float r2_kept = live ? r2_in[lo + tid] : -1.f;               // = cutoff_radius2(c), kept by the forward
                const uint32_t g = (uint32_t)key;
                if (g < (uint32_t)kp.P) {                                          // stale entries after an overflow are skipped
                    const float2 pg = xy[g];
                    const int rad = radii[g];
                    const uint32_t po = pair_off[g];
                    float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
                    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
                    if (live) { c = co[g]; c0 = rgb[3 * (size_t)g]; c1 = rgb[3 * (size_t)g + 1]; c2 = rgb[3 * (size_t)g + 2]; }
                    p = pg;
                    int x0, y0, x1, y1;
                    tile_rect(pg.x, pg.y, rad, kp.gx, kp.gy, x0, y0, x1, y1);
                    const int local = (ty - y0) * (x1 - x0) + (tx - x0);
                    s_pair[tid] = (tx >= x0 && tx < x1 && ty >= y0 && ty < y1) ? po + (uint32_t)local : 0xffffffffu;
                    if (live) {
                        const float4 q4 = scale_conic(c);
                        float2 *rec = reinterpret_cast<float2 *>(s_rec + tid * kEnt);
                        rec[0] = make_float2(q4.x, q4.y); rec[1] = make_float2(q4.z, q4.w);
                        rec[2] = make_float2(c0, c1);
                        rec[3] = make_float2(c2, __uint_as_float((uint32_t)(key >> 32)));
                        r2 = r2_kept;
                    }
                }
            }
            if (live) {
                reinterpret_cast<float2 *>(s_rec + tid * kEnt)[4] = p;
                s_r2[tid] = r2;
            }
        }
        __syncthreads();
        if (live) {
            // which of the staged splats can touch which of this wave's four sub-blocks: the forward's test, on the forward's numbers
            unsigned long long mt[4][kChunks];
#pragma unroll
            for (int c2 = 0; c2 < kChunks; c2++) {
                const int slot = (c2 << 6) + lane;
                unsigned long long mc[4] = { 0ull, 0ull, 0ull, 0ull };
                if ((c2 << 6) < cnt)                         // wave-uniform
                    wave_touch_masks(reinterpret_cast<const float2 *>(s_rec + slot * kEnt)[4], s_r2[slot], tx, ty, wave, mc);
#pragma unroll
                for (int r = 0; r < 4; r++) mt[r][c2] = mc[r];
            }
            int nsteps = 0, cnts[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                unsigned long long m[kChunks];
#pragma unroll
                for (int c2 = 0; c2 < kChunks; c2++) {
                    m[c2] = mt[r][c2];
                    // positions at or beyond the row's last contributor cannot matter: drop them from the mask
                    const uint32_t base = lo + ((uint32_t)c2 << 6);
                    if (row_max[r] <= base) m[c2] = 0;
                    else if (row_max[r] - base < 64u) m[c2] &= (1ull << (row_max[r] - base)) - 1ull;
                }
                cnts[r] = build_visit_list<kChunks, true, kEnt>(m, s_list[wave][r], lane);   // back to front
                nsteps = max(nsteps, cnts[r]);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) pad_visit_list(s_list[wave][r], cnts[r], nsteps, lane, (unsigned short)(kNull * kEnt));
            __builtin_amdgcn_wave_barrier();
            // Steps in which two rows of this wave hold the SAME splat (about one in five) must not do their slab updates in
            // one instruction; they are found here, 64 steps per pass, so that the replay only tests a scalar bit.
            unsigned long long conflict[kChunks];
#pragma unroll
            for (int c2 = 0; c2 < kChunks; c2++) {
                conflict[c2] = 0ull;
                if (!LAT && (c2 << 6) < nsteps) {
                    const int st = (c2 << 6) + lane;
                    const unsigned short *l0 = s_list[wave][0];
                    const uint32_t e0 = l0[st], e1 = l0[kListStride + st], e2 = l0[2 * kListStride + st], e3 = l0[3 * kListStride + st];
                    const uint32_t nul = (uint32_t)(kNull * kEnt);
                    const bool same = st < nsteps && ((e0 == e1 && e0 != nul) || (e0 == e2 && e0 != nul) || (e0 == e3 && e0 != nul) ||
                                                      (e1 == e2 && e1 != nul) || (e1 == e3 && e1 != nul) || (e2 == e3 && e2 != nul));
                    conflict[c2] = __ballot(same);
                }
            }
            unsigned long long conflict_s[kChunks];           // the same masks, pinned to scalar registers
#pragma unroll
            for (int c2 = 0; c2 < kChunks; c2++) conflict_s[c2] = uniform_u64(conflict[c2]);
            const unsigned short *list = s_list[wave][row];
            const unsigned char *rec_b = s_rec;
            // LAT: lanes that keep no sum write (zeros plus whatever) into distinct floats of the null splat's row of their slab
            unsigned char *slab = LAT ? reinterpret_cast<unsigned char *>(my_slot >= 0 ? &s_acc[wave * 4 + row][0][my_slot]
                                                                                      : &s_acc[wave * 4 + row][kNull][(lane & 15) % kAcc])
                                      : reinterpret_cast<unsigned char *>(&s_acc[wave][0][0] + (my_slot >= 0 ? my_slot : 0));
            const uint32_t slab_and = (!LAT || my_slot >= 0) ? 0xffffffffu : 0u;       // slot-less lanes of the latency build stay on their dummy float
            // entry of the first staged splat this pixel did NOT see in the forward pass (entries are slot * kEnt)
            const int lc_rel = (int)min(last_contributor - min(last_contributor, lo), (uint32_t)kBwdBatch) * kEnt;
            // The loop is arranged so that no LDS round trip sits between dependent instructions: the list entries of the
            // NEXT group are fetched while this group is processed, the colour records are fetched together with the
            // geometry records, and a step's slab value is read BEFORE its arithmetic and written back after it
            // (same wave, program order: the previous step's write is already ahead of the read in the LDS queue).
            uint2 pk = *reinterpret_cast<const uint2 *>(list);
            for (int k = 0; k < nsteps; k += 4) {
                const uint32_t ee[4] = { pk.x & 0xffffu, pk.x >> 16, pk.y & 0xffffu, pk.y >> 16 };
                pk = *reinterpret_cast<const uint2 *>(list + k + 4);          // the lists are padded: always readable
                v2f ds[4];
                float Gs[4], alphas[4];
                float4 cds[4];
                bool contribs[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {            // four independent evaluations (ILP)
                    const float2 *rec = reinterpret_cast<const float2 *>(rec_b + ee[u]);
                    const float2 q01 = rec[0], q23 = rec[1], c01 = rec[2], c23 = rec[3];
                    ds[u] = *reinterpret_cast<const v2f *>(rec + 4) - pix_f;
                    cds[u] = make_float4(c01.x, c01.y, c23.x, c23.y);
                    float p2;
                    eval_splat(make_float4(q01.x, q01.y, q23.x, q23.y), ds[u], p2, Gs[u], alphas[u]);
                    contribs[u] = (int)ee[u] < lc_rel && !(p2 > 0.0f) && !(alphas[u] < T4D_ALPHA_MIN);
                }
                const uint32_t cbits = (uint32_t)(conflict_s[kChunks == 1 ? 0 : (k >> 6)] >> (k & 63));
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const bool contrib = contribs[u];
                    const v2f d = ds[u];
                    const float G = Gs[u], alpha = alphas[u];
                    float *dst = reinterpret_cast<float *>(slab + (LAT ? (ee[u] & slab_and) : ee[u]));
                    const float old = *dst;              // early read of the slab value this step adds to
                    float e = 0.f, w = 0.f;
                    if (LAT) {
                        // the same operations in the same order as the exec-masked region below, on every lane; the selects keep
                        // the state of the lanes that do not contribute
                        const float4 cd = cds[u];
                        const float om = 1.f - alpha;
                        const float inv = __builtin_amdgcn_rcpf(om);
                        const float Tn = T * inv;
                        float q = fmaf(cd.x, dp01.x, fmaf(cd.y, dp01.y, cd.z * dp2));
                        if (DA) q = fmaf(cd.w, ddep, q) + dalp;
                        const float qma = q - acc;
                        const float dL_dalpha = qma * Tn;
                        T = contrib ? Tn : T;
                        w = contrib ? alpha * Tn : 0.f;
                        e = contrib ? G * dL_dalpha : 0.f;
                        acc = contrib ? fmaf(alpha, qma, acc) : acc;
                    } else if (contrib) {
                        // Per lane only what depends on the pixel: e = G * dL/dalpha and its first/second moments about
                        // the splat centre, and w * dL/dC.  Everything that is constant per splat (opacity, conic,
                        // 0.5*W, -0.5 ...) is applied ONCE per Gaussian after all tiles are summed (k_preprocess_bwd).
                        const float4 cd = cds[u];
                        const float om = 1.f - alpha;                              // >= 0.01
                        const float inv = __builtin_amdgcn_rcpf(om);          // (a Newton step on it: +1.9 % of the kernel, no decision depends on it - tools/experiments/README.md)
                        T = T * inv;
                        w = alpha * T;
                        float q = fmaf(cd.x, dp01.x, fmaf(cd.y, dp01.y, cd.z * dp2));
                        if (DA) q = fmaf(cd.w, ddep, q) + dalp;
                        const float qma = q - acc;                                 // acc = the colour behind THIS splat (background included)
                        const float dL_dalpha = qma * T;
                        e = G * dL_dalpha;
                        // ... and now behind the next one towards the eye: alpha q + (1 - alpha) acc as acc + alpha (q - acc), the
                        // difference being at hand (one instruction instead of two; upstream's two-product form rounds differently
                        // in the last bit)
                        acc = fmaf(alpha, qma, acc);
                    }
                    // lanes that do not contribute carry e = w = 0, so their ten products are exact zeros
                    const v2f ed = e * d, edd = ed * d, wdp = w * dp01;
                    float r[10] = { e, ed.x, ed.y, edd.x, ed.x * d.y, edd.y, wdp.x, wdp.y, w * dp2, DA ? w * ddep : 0.f };
                    const float tot = reduce10_row<!DA>(r);
                    // Plain read-add-write into the wave's slab (ds_add_f32 retires ~3 cycles per LANE on this part).  Idle
                    // rows add their zeros to the null splat's row, which nobody reads.
                    const bool add = my_slot >= 0;
                    if (LAT) {
                        *dst = old + tot;                // own slab per row: never a conflict; slot-less lanes hit their dummy float
                    } else if (!((cbits >> u) & 1u)) {
                        if (add) *dst = old + tot;
                    } else {
#pragma unroll
                        for (int rr = 0; rr < 4; rr++) {                 // two rows hold the same splat: one after the other
                            if (add && row == rr) *dst += tot;
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                        }
                    }
                }
            }
        }
        __syncthreads();
        // ---- write one record per pair (zeros when no wave touched it); fixed wave order => deterministic ----
        if (tid < cnt) {
            float a[10];
#pragma unroll
            for (int k = 0; k < 10; k++) a[k] = 0.f;
#pragma unroll
            for (int w = 0; w < kSlabs; w++) {
                float2 *src = reinterpret_cast<float2 *>(&s_acc[w][tid][0]);
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    const float2 b2 = src[k];
                    a[2 * k] += b2.x; a[2 * k + 1] += b2.y;
                    src[k] = make_float2(0.f, 0.f);                  // leave the slab zeroed for the next batch
                }
            }
            const uint32_t pr = s_pair[tid];
            if (pr < kp.cap) {
#pragma unroll
                for (int k = 0; k < kGP / 2; k++) grad_pair[(size_t)pr * (kGP / 2) + k] = make_float2(a[2 * k], a[2 * k + 1]);
            }
        }
        __syncthreads();
    }
    if (kp.tile_dot && seg_j == 0) {
        // The suffix recursion has reached the eye: acc = sum_i T_i alpha_i q_i + T_final bg . dL/dC = <colour, dL/dC> (+ <depth, dL/dD> +
        // <alpha, dL/dA>), this pixel's <outputs, cotangents> - the per-view sum costs one reduction per tile.
        // One float per wave, no barrier: a workgroup's lifetime is what this launch is made of.
        const float d = wave_sum_to_lane63(acc);
        if (lane == 63) kp.tile_dot[((size_t)v * kp.T + t_) * 4 + wave] = d;
    }
    }
}

