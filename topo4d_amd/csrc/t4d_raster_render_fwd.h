// t4d_raster_render_fwd.h - part of the translation unit t4d_raster.hip (included there, inside its anonymous namespace; not a
// stand-alone header).  A.3: pieces shared by both render kernels (alpha evaluation, sub-block touch masks, visit lists, empty-tile fill) and the forward render kernel.
// See t4d_raster.hip for the overview, the constants, the state layout and the kernel parameter block.
// ---------------------------------------------------------------------------------------------------------
// A.3 / A.4 shared pieces.
// Workgroup = 16x16 tile; wave w owns the 8x8 pixel block (w&1, w>>1); inside it DPP row r (16 lanes) owns the 4x4
// sub-block (r&1, r>>1) and lane i of the row the pixel (i&3, i>>2).  The four rows of a wave walk FOUR DIFFERENT visit
// lists at the same time (one per sub-block): a splat of Topo4D's size (cut-off radius ~5 px) touches a 4x4 sub-block
// 1.6x less often than an 8x8 block, so a wave needs that many fewer steps, every lane still sees its splats in list
// order (results are bit-identical to a per-pixel walk), and the backward's per-splat reduction runs over 16 lanes with
// row-local DPP only, for four splats at once.
// For every staged splat and sub-block a CONSERVATIVE test "can alpha reach 1/255 on any pixel centre of the sub-block?" decides
// whether the splat enters the sub-block's list: each wave tests the staged splats (one per lane, 64 at a time) against its
// own four sub-blocks and keeps the wave64 ballots as the bit masks its list builder walks (wave_touch_masks).  Skipped
// splats would have been rejected by the per-pixel alpha < 1/255 test anyway, so results are unchanged.
// ---------------------------------------------------------------------------------------------------------
constexpr float kLog2e = 1.4426950408889634f;


__device__ __forceinline__ void tile_pixel(int tid, int tx, int ty, int &px, int &py)
{
    const int w = tid >> 6, r = (tid >> 4) & 3, i = tid & 15;
    px = tx * T4D_TILE_X + ((w & 1) << 3) + ((r & 1) << 2) + (i & 3);
    py = ty * T4D_TILE_Y + ((w >> 1) << 3) + ((r >> 1) << 2) + (i >> 2);
}

// squared cut-off radius (pixels) beyond which opacity * exp(power) < 1/255 with margin; +inf = "cannot cull"
__device__ __forceinline__ float cutoff_radius2(const float4 co)
{
    const float lnarg = __logf(255.0f * co.w);               // alpha_max = opacity  =>  ln(255*opacity)
    if (!(lnarg > -1e-3f)) return -1.0f;                     // opacity < 1/255 (with margin): never contributes
    const float mid = 0.5f * (co.x + co.z);
    const float det = co.x * co.z - co.y * co.y;
    const float disc = mid * mid - det;
    // (hardware square root and reciprocals, ~1 ulp each: the margins below are four orders of magnitude wider, and the IEEE
    // sequences were 40 of this function's 60 instructions)
    const float lmin = det * __builtin_amdgcn_rcpf(mid + __builtin_amdgcn_sqrtf(fmaxf(disc, 0.f)));   // smallest eigenvalue of the conic, stable form
    if (!(lmin > 0.f) || !(mid > 0.f)) return __builtin_huge_valf();   // not positive definite / NaN: no culling
    return 2.0f * (lnarg + 2e-3f) * __builtin_amdgcn_rcpf(lmin) * 1.001f;
}

// Can the splat centred at p with squared cut-off r2 touch a 4x4 sub-block?  Asked for the FOUR sub-blocks of one wave (its DPP
// rows) at once, answered as wave masks: bit `lane` of out[r] <=> the splat whose centre and cut-off this LANE holds can touch
// row r of wave w of tile (tx, ty).  Every wave tests the staged splats against its own rows, 64
// splats per call, and gets the masks where the list builder wants them - in scalar registers; a staging wave computing all
// sixteen masks per splat, balloting them and handing them over through LDS cost the forward 190 vector instructions per
// wave and batch against 100 here (round 3).  r2 < 0 (a slot that holds no splat) touches nothing; the centre of such a
// slot must be finite.
__device__ __forceinline__ void wave_touch_masks(const float2 p, const float r2, const int tx, const int ty, const int w,
                                                 unsigned long long (&out)[4])
{
#pragma clang fp contract(off)
    float dx2[2], dy2[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const float x0 = (float)(tx * T4D_TILE_X + ((w & 1) << 3) + 4 * j), y0 = (float)(ty * T4D_TILE_Y + ((w >> 1) << 3) + 4 * j);
        const float ddx = fmaxf(fmaxf(x0 - p.x, p.x - (x0 + 3.f)), 0.f);
        const float ddy = fmaxf(fmaxf(y0 - p.y, p.y - (y0 + 3.f)), 0.f);
        dx2[j] = ddx * ddx; dy2[j] = ddy * ddy;
    }
#pragma unroll
    for (int r = 0; r < 4; r++) out[r] = __ballot(!(dx2[r & 1] + dy2[r >> 1] > r2));
}

// SGPR copy of lane `src_lane`'s value
__device__ __forceinline__ uint32_t lane_value(uint32_t v, int src_lane) { return __builtin_amdgcn_readlane(v, src_lane); }

typedef float v2f __attribute__((ext_vector_type(2)));      // packed-math pair (v_pk_*_f32 on gfx950)

// The ONE place alpha is evaluated, shared by forward and backward so that both take bit-identical decisions.
// q = (A, C, B, opacity) with (A, B, C) * log2(e) pre-multiplied by (-0.5, -1, -0.5) - note the ORDER: A and C are
// adjacent so that (A dx, C dy) is one packed multiply; d = splat centre - pixel.
// Returns p2 = power * log2(e), G = exp(power).
__device__ __forceinline__ void eval_splat(const float4 q, const v2f d, float &p2, float &G, float &alpha)
{
#pragma clang fp contract(off)
    const v2f qac = { q.x, q.y };
    const v2f m = qac * d;
    p2 = fmaf(fmaf(q.z, d.y, m.x), d.x, m.y * d.y);          // (A dx + B dy) dx + C dy^2: four instructions with the packed multiply
    G = __builtin_amdgcn_exp2f(p2);
    alpha = fminf(T4D_ALPHA_MAX, q.w * G);
}

__device__ __forceinline__ float4 scale_conic(const float4 co)      // (A, B, C, opacity) -> scaled (A, C, B, opacity)
{
#pragma clang fp contract(off)
    return make_float4(co.x * (-0.5f * kLog2e), co.z * (-0.5f * kLog2e), co.y * (-kLog2e), co.w);
}

__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v)
{
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// Expand a 64-bit-per-chunk visit mask into a compact list of staged-splat entries (ascending when !REVERSE,
// descending when REVERSE).  An entry is the splat's slot in the staging arrays times SCALE, i.e. directly the byte
// offset the consumer needs, so the hot loops spend no instructions on address arithmetic.  The hot loops are plain
// counted loops: almost no scalar-unit work per splat (the CU's single scalar unit is what bounded the first version
// of these kernels).
template <int NCHUNK, bool REVERSE, int SCALE>
__device__ __forceinline__ int build_visit_list(const unsigned long long (&m)[NCHUNK], unsigned short *list, const int lane,
                                                const int chunk0 = 0)      // chunk0: staged slot of m[0]'s first bit, in chunks of 64
{
    int cnt = 0;
#pragma unroll
    for (int cc = 0; cc < NCHUNK; cc++) {
        const int c = REVERSE ? NCHUNK - 1 - cc : cc;
        const unsigned long long mw = m[c];
        if (mw == 0ull) continue;                               // wave-uniform: most chunks of most batches are empty
        const int below = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mw >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mw, 0u));
        const int tot = __builtin_popcountll(mw);
        // the mask is wave-uniform: it becomes the exec mask of the store as it is (a per-lane bit test cost three instructions)
        if (__builtin_amdgcn_inverse_ballot_w64(mw)) list[cnt + (REVERSE ? tot - 1 - below : below)] = (unsigned short)((((c + chunk0) << 6) + lane) * SCALE);
        cnt += tot;
    }
    return cnt;
}

// pad a row's list with the null entry up to (and GROUP - 1 entries beyond) the wave's longest list: every row then walks
// the same number of steps, GROUP at a time, without a per-step bounds test
template <int GROUP = 4>
__device__ __forceinline__ void pad_visit_list(unsigned short *list, const int cnt, const int nsteps, const int lane,
                                               const unsigned short null_entry)
{
#pragma clang loop vectorize(disable) unroll(disable)
    for (int p2 = cnt + lane; p2 < nsteps + GROUP - 1; p2 += 64) list[p2] = null_entry;
}

// The pixels of EMPTY tiles (config 4: two thirds of 2048^2): background colour, zero depth, zero alpha - 20 bytes per pixel that
// no splat ever touches.  Written tile by tile (a tile's row is 64 bytes of a plane, a wave's store 32) they went to HBM at
// 2.5 TB/s and made up a third of k_render_fwd at config 4 (803 us for an all-empty launch).  Here one workgroup takes a whole
// ROW of tiles of a view and walks it in image order, 16 bytes per lane, skipping the tiles that hold splats: neighbouring
// empty tiles become one long contiguous store per image row.  These workgroups are spread evenly between the tile
// workgroups of the same launch (k_render_fwd): bandwidth work next to issue-bound work.
__device__ __forceinline__ void fill_empty_tile_row(const KP &kp, const uint32_t j)
{
    const int tid = threadIdx.x;
    const int v = (int)(j / (uint32_t)kp.gy), ty = (int)(j - (uint32_t)v * (uint32_t)kp.gy);
    const uint32_t *tc = kp.tile_count + (size_t)v * kp.T + (size_t)ty * kp.gx;
    const float *vr = kp.views + (size_t)v * T4D_VIEW_FLOATS;
    const float b0 = vr[35], b1 = vr[36], b2 = vr[37];
    const int y0 = ty * T4D_TILE_Y, rows = min(T4D_TILE_Y, kp.H - y0);
    const size_t HW = (size_t)kp.H * kp.W;
    float *oc = kp.out_color + (size_t)v * 3 * HW + (size_t)y0 * kp.W;
    float *od = kp.out_depth + (size_t)v * HW + (size_t)y0 * kp.W;
    float *oa = kp.out_alpha + (size_t)v * HW + (size_t)y0 * kp.W;
    if (kp.fill_vec) {
        const int qw = kp.W >> 2;                        // 16-byte groups per image row; four of them per tile
        int r = tid / qw, q = tid - r * qw;
        const int dr = kBlock / qw, dq = kBlock - dr * qw;
        while (r < rows) {
            if (tc[q >> 2] == 0u) {
                const size_t o = (size_t)r * kp.W + 4 * q;
                *reinterpret_cast<float4 *>(oc + o) = make_float4(b0, b0, b0, b0);
                *reinterpret_cast<float4 *>(oc + HW + o) = make_float4(b1, b1, b1, b1);
                *reinterpret_cast<float4 *>(oc + 2 * HW + o) = make_float4(b2, b2, b2, b2);
                *reinterpret_cast<float4 *>(od + o) = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4 *>(oa + o) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            r += dr; q += dq;
            if (q >= qw) { q -= qw; r++; }
        }
    } else {
        int r = tid / kp.W, x = tid - r * kp.W;
        const int dr = kBlock / kp.W, dx = kBlock - dr * kp.W;
        while (r < rows) {
            if (tc[x / T4D_TILE_X] == 0u) {
                const size_t o = (size_t)r * kp.W + x;
                oc[o] = b0; oc[HW + o] = b1; oc[2 * HW + o] = b2; od[o] = 0.f; oa[o] = 0.f;
            }
            r += dr; x += dx;
            if (x >= kp.W) { x -= kp.W; r++; }
        }
    }
}

#ifndef T4D_FWD_WAVES
#define T4D_FWD_WAVES 7          // 72 VGPRs (round 3, after the staging part shrank: 6 waves 105.6 us, 7 waves 101.6 us at config 2; 8 waves spill: 117 us)
#endif
// Two instantiations of each per-tile render kernel.  LAT = false is the THROUGHPUT build (many tiles in flight, bound by
// vector-ALU issue: registers are capped for occupancy, steps go four at a time).  LAT = true is the LATENCY build, chosen by
// the host when a launch holds too few tiles to fill the chip (the reference's own call shape: ONE view of 768 tiles per
// call, train.py:661-673): every CU then runs one workgroup whose duration is the dependent-instruction chain of its
// longest visit list, so this build spends registers and LDS freely on instruction-level parallelism - eight steps per
// group with all their LDS reads issued up front, no exec-mask branches between the steps, one backward slab per DPP row
// (no same-splat conflicts to serialise).  Per-pixel arithmetic and its order are IDENTICAL in both builds: forward
// outputs are bit-equal; the backward's partial sums are added up in a different (still fixed) order.
#ifndef T4D_LAT_WAVES
#define T4D_LAT_WAVES 2          // most waves per SIMD the latency build is compiled for (register budget 512 / this)
#endif
#define T4D_FWD_ATTR __attribute__((amdgpu_waves_per_eu(LAT ? 1 : T4D_FWD_WAVES, LAT ? T4D_LAT_WAVES : T4D_FWD_WAVES)))
// FB: splats staged per batch.  SEGN != 0: the launch is small enough for the segmented backward: visit lists are built and walked
// per SEGN list positions (128, or 64 for a one-view launch: seg_positions), and the blend state at every such boundary is kept for
// the backward (write_snapshot).
// PRUNE: finished sub-blocks walk empty lists (below).  A template parameter because its mere presence costs the 72-register
// build 2.5 % at config 2 (register allocation, not executed instructions: a run-time gate that is never true costs the same),
// where no list is long enough for it to matter: the host instantiates it for launches that may hold long lists.
// LONG: a big one-view launch whose long tiles (>= kp.seg_min_pairs pairs) are rendered by the depth-parallel kernels of
// t4d_raster_render_fwd_long.h: this launch leaves them out.
template <bool LAT, int FB, int SEGN, bool PRUNE, bool LONG = false>
__global__ __launch_bounds__(kBlock) T4D_FWD_ATTR void k_render_fwd(const KP kp)
{
    constexpr bool SEG = SEGN != 0;
    constexpr int kSeg = SEG ? SEGN : 128;           // (shadows the global default: this launch's segment length)
    constexpr int kU = LAT ? 8 : 4;                  // steps per group
    // splats staged per batch: the latency build has the LDS of a whole CU and lives as long as its longest tile - fewer batches
    constexpr int kFB = FB;
    constexpr int kNull = kFB;                 // staged slot that can never contribute (opacity 0)
    constexpr int kSub = SEG ? kSeg : kFB;           // list positions per visit-list round
    constexpr int kSubChunks = kSub / 64, kNSub = kFB / kSub;
    constexpr int kListStride = kSub + 8;      // u16 entries per row list (multiple of 4: 8-byte aligned rows)
    constexpr int kRec = 48;                         // bytes per staged splat: xy, cut-off r2 (12, +4 pad) | scaled conic + opacity | rgb + depth
    static_assert(kFB <= kBlock && kFB % 64 == 0 && kFB % kSub == 0, "one staging thread per slot (it clears the slot when the list is shorter)");
    // latency build: the workgroup sorts its own tile's bin first (one launch and one trip through memory less than
    // k_sort_tiles -> k_render_fwd) and stages from the sorted keys it still holds: a sort buffer of its own.
    // FUSE (the one-view dense pass, LONG): the workgroup sorts the bins of ALL its tiles before it renders any of them, through
    // the staging area itself (records and lists are not in use yet: the CU keeps its seven workgroups).  Such a launch is one frame
    // at a time by nature (Topo4D's texture loop), so k_sort_tiles - 44 us with nothing else on the chip - cannot hide in another
    // frame's gaps; here a workgroup's sorting runs beside the other workgroups' blending.  For launches with frames in flight the
    // same fusion measured 1.7 % SLOWER (tools/experiments/README.md, round 6): they keep k_sort_tiles, and their instantiations of
    // this kernel are untouched by the `if constexpr` below.
    constexpr bool FUSE = LONG;
    constexpr int kRecBytes = ((kFB + 1) * kRec + 15) / 16 * 16;
    unsigned char *s_rec;
    unsigned short (*s_list)[4][kListStride];
    unsigned long long *s_sort;
    if constexpr (FUSE) {
        constexpr int kStage = kRecBytes + 4 * 4 * kListStride * 2;
        constexpr int kRaw = kStage < kSortLdsCap * 8 ? kSortLdsCap * 8 : kStage;
        __shared__ __attribute__((aligned(16))) unsigned char s_raw[kRaw];
        s_rec = s_raw;
        s_list = reinterpret_cast<unsigned short (*)[4][kListStride]>(s_raw + kRecBytes);
        s_sort = reinterpret_cast<unsigned long long *>(s_raw);
    } else {
        __shared__ __attribute__((aligned(16))) unsigned char s_rec_[(kFB + 1) * kRec];
        __shared__ __attribute__((aligned(8))) unsigned short s_list_[4][4][kListStride];
        __shared__ unsigned long long s_sort_[LAT ? kSortLdsCap : 1];
        s_rec = s_rec_; s_list = s_list_; s_sort = s_sort_;
    }
    __shared__ uint32_t s_wave_done[4];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, row = lane >> 4;
    // fill workgroups are spread evenly over the launch: workgroup b is one iff floor(b F / total) steps up at b
    const uint32_t total_blocks = kp.tile_blocks + kp.fill_blocks;
    const uint32_t fills_before = (uint32_t)(((unsigned long long)blockIdx.x * kp.fill_blocks) / total_blocks);
    if ((uint32_t)(((unsigned long long)(blockIdx.x + 1u) * kp.fill_blocks) / total_blocks) != fills_before) {
        fill_empty_tile_row(kp, fills_before);
        return;
    }
    if constexpr (FUSE) {
        if (kp.fused_sort) {
            // (in a loop of its own: inside the tile loop the sort's registers lived beside the walk's - 17 vector and 103 scalar
            // spills under the 72-register budget.  Also the bin of a tile the depth-parallel kernels render - exactly kSortLdsCap
            // pairs: the long-bin kernels take only longer ones)
            for (uint32_t item = blockIdx.x - fills_before; item < (uint32_t)(kp.V * kp.T); item += kp.tile_blocks) {
                const uint4 it = kp.items[item];
                if (it.w == 0u) break;
                if (it.z > 1u) {
                    sort_one_bin<false, kBlock>(kp, (int)(it.x >> 20), it.y, it.z, s_sort, tid, wave, lane);
                    __syncthreads();                 // the sort buffer is reused by the next bin
                }
            }
            __threadfence_block();                   // the sorted keys are staged from memory below, by other threads of this workgroup
            __syncthreads();
        }
    }
    if (tid < kRec / 4) reinterpret_cast<float *>(s_rec + kNull * kRec)[tid] = 0.f;
    for (uint32_t item = blockIdx.x - fills_before; item < (uint32_t)(kp.V * kp.T); item += kp.tile_blocks) {
    const uint4 it = kp.items[item];
    if (it.w == 0u) break;                           // ordered by length: only empty tiles remain, and those are not ours
    const int v = (int)(it.x >> 20), t_ = (int)(it.x & 0xfffffu);
    const int ty = t_ / kp.gx, tx = t_ - ty * kp.gx;
    const uint32_t off = it.y, n = it.z;
    if (LONG && n >= kp.seg_min_pairs) continue;     // workgroup-uniform, before anything of this tile is touched
    const unsigned long long *keys = kp.keys + (size_t)v * kp.cap + off;
    float *r2_out = kp.cut_r2 + (size_t)v * kp.cap + off;
    const float2 *xy = kp.xy + (size_t)v * kp.P;
    const float4 *co = kp.conic_opacity + (size_t)v * kp.P;
    const float *rgb = kp.shs ? kp.rgb + (size_t)v * kp.P * 3 : kp.colors_precomp + 3 * param_row0(kp, v);
    // segmented backward: this tile's snapshot slots (a tile of one segment keeps none: its replay starts at the list's end)
    float *snap = nullptr;
    if (SEG && n > (uint32_t)kSeg && n >= kp.seg_min_pairs)
        snap = kp.snap + ((size_t)v * kp.slots_per_view + seg_slot0(kp, off, (uint32_t)t_)) * (kSnapFloats * kBlock) + tid;

    int px, py;
    tile_pixel(tid, tx, ty, px, py);
    const bool inside = px < kp.W && py < kp.H;
    const v2f pix_f = { (float)px, (float)py };
    // The background colour is fetched HERE, into scalar registers.  Fetched where it is used - between the output stores - each
    // of its three loads was followed by a wait for ALL outstanding memory operations (gfx9 counts loads and stores in one
    // counter): store, wait for it, load, wait, store ... three dependent round trips at the end of every tile (config 4:
    // 1,143 -> 1,111 us).
    const float *vr = kp.views + (size_t)v * T4D_VIEW_FLOATS;
    const float bg0 = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(vr[35])));
    const float bg1 = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(vr[36])));
    const float bg2 = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(vr[37])));
    unsigned long long done_m = __ballot(!inside);   // pixels that take no more splats, as a wave mask
    uint32_t gate = inside ? 0xffffffffu : 0u;       // (latency build: the same per lane, all ones while the pixel takes splats)
    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f;
    uint32_t last_contributor = 0;

    // ---- latency build: sort, then keep one batch of records and two batches of keys in flight ----
    // A lone workgroup per CU lives through every memory round trip of its tile: as the kernel was written a batch began with
    // key -> (centre, conic, colour), two dependent trips while all four waves waited.  Here the records of batch b + 1 are
    // requested before the walk of batch b and the keys of batch b + 2 with them; only the first batch waits.
    bool keys_lds = false;
    unsigned long long k_cur = ~0ull, k_nxt = ~0ull;
    float2 pre_p = make_float2(0.f, 0.f);
    float4 pre_c = make_float4(0.f, 0.f, 0.f, 0.f);
    float pre_r0 = 0.f, pre_r1 = 0.f, pre_r2 = 0.f;
    if (LAT) {
        static_assert(!LAT || kFB == kBlock, "the latency build stages one splat per thread");
        if (kp.fused_sort) {
            sort_one_bin<true, kBlock>(kp, v, off, n, s_sort, tid, wave, lane);
            __threadfence_block();
            __syncthreads();
            keys_lds = n <= (uint32_t)kSortLdsCap;
        }
        if ((uint32_t)tid < n) k_cur = keys_lds ? s_sort[tid] : keys[tid];
        if ((uint32_t)(kFB + tid) < n) k_nxt = keys_lds ? s_sort[kFB + tid] : keys[kFB + tid];
        const uint32_t g0 = (uint32_t)k_cur;
        if (g0 < (uint32_t)kp.P) {
            pre_p = xy[g0]; pre_c = co[g0];
            pre_r0 = rgb[3 * (size_t)g0]; pre_r1 = rgb[3 * (size_t)g0 + 1]; pre_r2 = rgb[3 * (size_t)g0 + 2];
        }
    }

    for (uint32_t b = 0; b < n; b += kFB) {
        if (b != 0) {                                // a further batch: needed only while some pixel of the tile is unfinished
            if (lane == 0) s_wave_done[wave] = done_m == ~0ull ? 1u : 0u;
            __syncthreads();                         // (also: everyone has left the previous batch's records)
            if ((s_wave_done[0] & s_wave_done[1] & s_wave_done[2] & s_wave_done[3]) != 0u) break;
        }
        if (LAT) {
            float4 head = make_float4(0.f, 0.f, -1.f, 0.f);      // (x, y, cut-off r2, -): a slot without a splat touches nothing
            if (b + tid < n) {
                // (g >= P: a stale entry of a truncated list - lazy mode after an arena overflow - is ignored)
                if ((uint32_t)k_cur < (uint32_t)kp.P) {
                    unsigned char *rec = s_rec + tid * kRec;
                    head = make_float4(pre_p.x, pre_p.y, cutoff_radius2(pre_c), 0.f);
                    *reinterpret_cast<float4 *>(rec + 16) = scale_conic(pre_c);
                    *reinterpret_cast<float4 *>(rec + 32) = make_float4(pre_r0, pre_r1, pre_r2, __uint_as_float((uint32_t)(k_cur >> 32)));
                }
                r2_out[b + tid] = head.z;                // the backward stages the same splats: it reads the cut-off back
            }
            *reinterpret_cast<float4 *>(s_rec + tid * kRec) = head;
        } else if (kFB == kBlock || tid < kFB) {
            float4 head = make_float4(0.f, 0.f, -1.f, 0.f);      // (x, y, cut-off r2, -): a slot without a splat touches nothing
            if (b + tid < n) {
                const unsigned long long key = keys[b + tid];
                const uint32_t g = (uint32_t)key;
                // g >= P only happens in lazy mode after an arena overflow (slots of dropped pairs hold stale bytes):
                // such entries are ignored instead of being dereferenced
                if (g < (uint32_t)kp.P) {
                    const float2 p = xy[g];
                    const float4 c = co[g];
                    unsigned char *rec = s_rec + tid * kRec;
                    head = make_float4(p.x, p.y, cutoff_radius2(c), 0.f);
                    *reinterpret_cast<float4 *>(rec + 16) = scale_conic(c);
                    *reinterpret_cast<float4 *>(rec + 32) = make_float4(rgb[3 * (size_t)g], rgb[3 * (size_t)g + 1], rgb[3 * (size_t)g + 2],
                                                                       __uint_as_float((uint32_t)(key >> 32)));
                }
                r2_out[b + tid] = head.z;                // the backward stages the same splats: it reads the cut-off back
            }
            *reinterpret_cast<float4 *>(s_rec + tid * kRec) = head;
        }
        __syncthreads();
        if (LAT) {
            // the next batch's records and the keys of the one after it: in flight during this batch's walk.  Requested BEHIND the
            // barrier (a barrier waits for every outstanding memory operation of the wave), by every thread, finished wave or not.
            k_cur = k_nxt;
            k_nxt = ~0ull;
            const uint32_t pos2 = b + 2u * kFB + (uint32_t)tid;
            if (pos2 < n) k_nxt = keys_lds ? s_sort[pos2] : keys[pos2];
            const uint32_t g1 = (uint32_t)k_cur;
            if (g1 < (uint32_t)kp.P) {
                pre_p = xy[g1]; pre_c = co[g1];
                pre_r0 = rgb[3 * (size_t)g1]; pre_r1 = rgb[3 * (size_t)g1 + 1]; pre_r2 = rgb[3 * (size_t)g1 + 2];
            }
        }
        if (done_m == ~0ull) continue;               // wave-uniform; still takes part in the barriers above
        uint32_t last_e = 0xffffffffu;               // entry of the last splat blended in this batch
#pragma clang loop unroll(disable)
        for (int sub = 0; sub < kNSub; sub++) {      // (one round per batch unless SEG)
        const uint32_t sub_lo = b + (uint32_t)(sub * kSub);
        if (sub != 0 && !(sub_lo < n)) break;
        // A sub-block whose sixteen pixels have all finished takes no more splats: its row walks an empty list, and the wave steps
        // as often as the longest list of the rows that still blend (silhouette tiles of a dense pass hold thousands of pairs
        // and a handful of unsaturated pixels: one view of 10^6 Gaussians 642 -> 298 us).  Scalar work is scarce (one scalar
        // unit per CU): the question is asked once per round, and only where a pixel of the wave has finished at all.
        uint32_t rows_done = 0u;
        if (PRUNE && done_m != 0ull) {
#pragma unroll
            for (int r = 0; r < 4; r++) rows_done |= (((done_m >> (16 * r)) & 0xffffull) == 0xffffull ? 1u : 0u) << r;
        }
        // which of the staged splats can touch which of this wave's four sub-blocks (= DPP rows)
        unsigned long long m[4][kSubChunks];
#pragma unroll
        for (int c4 = 0; c4 < kSubChunks; c4++) {
            unsigned long long mc[4] = { 0ull, 0ull, 0ull, 0ull };
            if (sub_lo + ((uint32_t)c4 << 6) < n) {      // wave-uniform: short lists leave most chunks of a batch empty
                const float4 head = *reinterpret_cast<const float4 *>(s_rec + (((sub * kSubChunks + c4) << 6) + lane) * kRec);
                wave_touch_masks(make_float2(head.x, head.y), head.z, tx, ty, wave, mc);
                if (PRUNE && rows_done != 0u) {
#pragma unroll
                    for (int r = 0; r < 4; r++) mc[r] = ((rows_done >> r) & 1u) ? 0ull : mc[r];
                }
            }
#pragma unroll
            for (int r = 0; r < 4; r++) m[r][c4] = mc[r];
        }
        int nsteps = 0, cnts[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {                // one visit list per sub-block
            cnts[r] = build_visit_list<kSubChunks, false, kRec>(m[r], s_list[wave][r], lane, sub * kSubChunks);
            nsteps = max(nsteps, cnts[r]);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) pad_visit_list<kU>(s_list[wave][r], cnts[r], nsteps, lane, (unsigned short)(kNull * kRec));
        __builtin_amdgcn_wave_barrier();
        const unsigned short *list = s_list[wave][row];
        for (int k = 0; k < nsteps; k += kU) {
            uint32_t e[kU];
#pragma unroll
            for (int h = 0; h < kU / 4; h++) {
                const uint2 pk = *reinterpret_cast<const uint2 *>(list + k + 4 * h);
                e[4 * h] = pk.x & 0xffffu; e[4 * h + 1] = pk.x >> 16; e[4 * h + 2] = pk.y & 0xffffu; e[4 * h + 3] = pk.y >> 16;
            }
            float alpha[kU];
            unsigned long long valid[kU];                // lane predicates are kept as wave masks: see the blend below
            float4 cds[kU];
#pragma unroll
            for (int u = 0; u < kU; u++) {               // independent evaluations: ILP hides LDS / exp latency
                const v2f g_xy = *reinterpret_cast<const v2f *>(s_rec + e[u]);
                if (LAT) cds[u] = *reinterpret_cast<const float4 *>(s_rec + e[u] + 32);      // every LDS read of the group up front
                float p2, G;
                eval_splat(*reinterpret_cast<const float4 *>(s_rec + e[u] + 16), g_xy - pix_f, p2, G, alpha[u]);
                if (LAT) {                               // the two rejections fold into alpha itself (see the blend below)
                    const float a1 = p2 > 0.0f ? 0.f : alpha[u];
                    alpha[u] = a1 < T4D_ALPHA_MIN ? 0.f : a1;
                } else {
                    valid[u] = __ballot(!(p2 > 0.0f)) & __ballot(!(alpha[u] < T4D_ALPHA_MIN));
                }
            }
            // (no "does any lane blend?" test: with four different splats in flight per step the answer is almost always yes)
            if (LAT) {
                // The latency build's blend: ONE wave per SIMD walks a dependent chain, so what counts is the LENGTH of the chain from
                // one splat's transmittance to the next, not the instruction count.  With wave masks that chain crosses from the
                // vector to the scalar unit and back per splat (compare -> mask logic -> select: ~125 cycles per step measured);
                // here it stays in the vector unit: a splat that must not blend - rejected, or its pixel finished (gate = 0) - takes
                // part with alpha = 0, for which every update below is the identity (T * 1, C + c * 0), bit for bit what the
                // throughput build's skipped update leaves.  (T >= T_STOP holds for every pixel that still takes splats, so a
                // zero alpha can never raise `stop`.)
#pragma unroll
                for (int u = 0; u < kU; u++) {
                    const float a = __uint_as_float(__float_as_uint(alpha[u]) & gate);
                    const float test_T = T * (1.f - a);
                    const bool stop = test_T < T4D_T_STOP;
                    const float w = stop ? 0.f : a * T;
                    gate = stop ? 0u : gate;
                    const float4 cd = cds[u];
                    C0 = fmaf(cd.x, w, C0); C1 = fmaf(cd.y, w, C1); C2 = fmaf(cd.z, w, C2);
                    D = fmaf(cd.w, w, D);
                    T = stop ? T : test_T;
                    last_e = w != 0.f ? e[u] : last_e;
                }
                done_m = __ballot(gate == 0u);
                if (done_m == ~0ull) break;
                continue;
            }
#pragma unroll
            for (int u = 0; u < kU; u++) {               // blending is sequential in list order
                // Predicates as 64-bit wave masks combined with scalar instructions: written with bools, the compiler evaluates
                // "below" and "not below" as two vector compares (one instruction in 25 per step).
                const float test_T = T * (1.f - alpha[u]);
                const unsigned long long below = __ballot(test_T < T4D_T_STOP);
                const unsigned long long live = valid[u] & ~done_m;
                done_m |= live & below;
                const bool ok = __builtin_amdgcn_inverse_ballot_w64(live & ~below);
                const float4 cd = LAT ? cds[u] : *reinterpret_cast<const float4 *>(s_rec + e[u] + 32);
                const float w = ok ? alpha[u] * T : 0.f;
                C0 = fmaf(cd.x, w, C0); C1 = fmaf(cd.y, w, C1); C2 = fmaf(cd.z, w, C2);
                D = fmaf(cd.w, w, D);
                T = ok ? test_T : T;
                last_e = ok ? e[u] : last_e;
            }
            if (done_m == ~0ull) break;
        }
        if (SEG) {
            // the blend state in front of list position sub_lo + kSeg, for the backward segment that ends there.  A pixel that is
            // finished keeps its final state, which the backward takes from the final snapshot: a finished WAVE writes nothing.
            if (done_m == ~0ull) break;
            if (snap != nullptr && sub_lo + (uint32_t)kSeg < n) {
                float *sp = snap + (size_t)(sub_lo / kSeg) * (kSnapFloats * kBlock);
                sp[0] = T; sp[kBlock] = C0; sp[2 * kBlock] = C1; sp[3 * kBlock] = C2; sp[4 * kBlock] = D;
            }
        }
        }
        if (last_e != 0xffffffffu) last_contributor = b + ((last_e * 43691u) >> 21) + 1u;     // entry / 48 for entries < 2^17
    }
    if (SEG && snap != nullptr) {                     // the final state, in the tile's last slot
        float *sp = snap + (size_t)((n - 1u) / kSeg) * (kSnapFloats * kBlock);
        sp[0] = T; sp[kBlock] = C0; sp[2 * kBlock] = C1; sp[3 * kBlock] = C2; sp[4 * kBlock] = D;
    }
    if (inside) {
        const size_t HW = (size_t)kp.H * kp.W, pix = (size_t)py * kp.W + px;
        if (n != 0) {                                 // the backward never visits an empty tile: no replay state for it
            kp.final_T[(size_t)v * HW + pix] = T;
            kp.n_contrib[(size_t)v * HW + pix] = last_contributor;
        }
        float *oc = kp.out_color + (size_t)v * 3 * HW;
        oc[pix] = C0 + T * bg0;
        oc[HW + pix] = C1 + T * bg1;
        oc[2 * HW + pix] = C2 + T * bg2;
        kp.out_depth[(size_t)v * HW + pix] = D;
        // alpha = sum of the blend weights w_i = T_i - T_(i+1): the sum telescopes to 1 - T_final, which is at hand (upstream adds
        // the weights up one by one; one add per step less here, and one rounding instead of one per splat)
        kp.out_alpha[(size_t)v * HW + pix] = 1.f - T;
    }
    __syncthreads();                                 // staging buffers are reused by the next tile
    }
}

