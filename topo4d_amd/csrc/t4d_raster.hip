// t4d_raster.hip — MI355X (gfx950 / CDNA4) differentiable Gaussian-splatting rasterizer + C ABI.
//
// Replaces, for Topo4D, the un-vendored CUDA package `diff_gaussian_rasterization` that the reference calls at
// train.py:307,388,463,484 (boundary: helpers.py:63-112).  Written from the published algorithm (SURVEY.md
// Appendix A) for wave64 / LDS / 8-XCD hardware; it is not a translation of the CUDA sources (which are not
// even present under /root/reference).  Differences in STRUCTURE from upstream, all result-preserving:
//   * V views of the same Gaussians go through one set of launches (the view is part of every work item);
//   * binning is count (LDS tile histogram per workgroup, one global atomic per touched tile, the rank of every pair
//     remembered) -> per-view tile scan -> atomic-free scatter -> per-tile sort on the 64-bit key
//     (depth bits << 32 | Gaussian index).  That reproduces upstream's order (stable radix sort on tile|depth of
//     pairs emitted in index order) without a global sort and without a host round trip;
//   * tiles are processed in descending list length (work items built on the device), which balances the 8 XCDs;
//   * inside a tile every 16-lane DPP row owns a 4x4 pixel sub-block with its own culled visit list, so the four rows of
//     a wave work on four different splats at a time;
//   * the backward uses no atomics at all: ten sums per (row,splat) are reduced with bank-masked DPP adds, added to
//     per-wave LDS slabs by plain read-add-write, summed over the waves in fixed order into one raw-moment record per
//     (Gaussian,tile) pair, and the per-Gaussian kernel gathers its pairs in fixed order.  Gradients are
//     bit-reproducible run to run.
//
// Kernels (DESIGN.md has the bytes/roofline of each):
//   k_preprocess      A.1  per (view,Gaussian): cull, project, cov3D, EWA cov2D, conic, radius, tile rect, SH colour;
//                          + per-tile counts and pair ranks + pair-slot allocation (one returning atomic per workgroup)
//   k_tile_chunk_sums, k_scan_tiles  A.2  per (view, 1024-tile chunk): exclusive scan of tile counts -> tile offsets, overflow
//                          status, longest list, length buckets (the first kernel only when a view has more than one chunk)
//   k_scatter         A.2  per (view,Gaussian): key -> tile_off + rank (no atomics); tail blocks build the work items
//   k_sort_tiles      A.2  per work item: sort the bin by (depth bits, index): runs of 64 sorted in registers, merged by
//                          ranking (<= 512 keys: one pass; <= 2048: log levels in LDS)
//   k_sort_long       A.2  bins beyond 2048 keys (dense passes): a CU each - 1024 threads, 128 KiB of LDS for up to 16,384 keys,
//                          longer ones chunk-sorted and merged in global memory
//   k_render_fwd      A.3  per work item: 256 threads = 4 wave64 = 16 DPP rows, one 4x4 sub-block each; front-to-back blend;
//                          empty tiles are written by row-fill workgroups of the same launch (fill_empty_tile_row)
//   k_render_bwd      A.4  per work item: back-to-front replay, row-local reduction, one record per pair
//                          (both render kernels exist in a throughput and a latency build: see k_render_fwd)
//   k_preprocess_bwd  A.5  per (view,Gaussian): gather pair records, conic/cov2D/projection/cov3D chain rule
//   k_sh_bwd16, k_sh_bwd   SH colours: dL/dshs and the view-direction term of dL/dmeans3D (degree 3 / any degree)
//   k_view_dot_*, k_mark_visible: small utilities of the ABI
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "../../include/t4d_config.h"
#include "../../include/topo4d_raster.h"

#define T4D_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

// T4D_ABL (ablation builds, tools/ablate.sh; never defined in the shipped library) - whole phases only, nothing inside the step
// bodies of the render kernels (the in-step ablations and the s_memtime stamps of rounds 2-3 are recorded, with their numbers, in
// tools/experiments/README.md):
//   3 = backward: skip the whole visit loop (staging + write-out only) 4 = forward: skip blending (alpha evaluation only)
//   5 = forward: skip the whole visit loop      6 / 7 = preprocess: stop before / after the pair-slot allocation
#ifndef T4D_ABL
#define T4D_ABL 0
#endif
constexpr int kBlock = 256;          // threads per workgroup everywhere (4 wave64)
#ifndef T4D_FWD_BATCH
#define T4D_FWD_BATCH 192
#endif
constexpr int kFwdBatch = T4D_FWD_BATCH;   // splats staged in LDS per round of the forward blend (one per thread: <= 256; sweep of round 3 with 7 waves per SIMD, config 2 / config 4: 128: 106 / 1,197 us, 192: 101.5 / 1,113, 256: 102 / 1,134)
#ifndef T4D_BWD_BATCH
#define T4D_BWD_BATCH 128
#endif
constexpr int kBwdBatch = T4D_BWD_BATCH;   // splats staged per round of the backward replay (64 or 128)
constexpr int kSortLdsCap = 2048;    // keys sorted in LDS by k_sort_tiles (16 KiB: 8 workgroups per CU); longer bins go to k_sort_long
constexpr int kLongBlock = 1024;     // k_sort_long: threads per workgroup ...
constexpr int kLongCap = 16384;      // ... and keys it sorts in LDS (128 KiB: one workgroup per CU)
constexpr int kScanChunk = 1024;     // tiles scanned per workgroup of k_scan_tiles
constexpr int kRankSortMax = 512;    // bins up to this length: register-sorted runs of 64 + one ranking pass (one barrier)
constexpr int kHist = 1024;          // per-workgroup LDS tile histogram (bounding box of the tiles a workgroup touches)
constexpr int kBuckets = 24;         // tile-length classes (floor(log2 n), descending; last = empty) for launch ordering
constexpr int kGP = T4D_GRAD_PAIR_FLOATS;
constexpr int kCursorSegs = 8;       // pair-slot cursors per view (same-address returning atomics are serial: see k_preprocess)
// Small launches (the reference's own call shape: ONE view per call, train.py:661-673; a view-sharded rank: three views) cannot
// fill the chip with whole tiles: a kernel lasts as long as its LONGEST tile list is walked by one workgroup.  For launches of
// at most kSegMaxTiles tiles the backward is therefore cut along DEPTH: a tile list of n pairs becomes ceil(n / kSeg)
// independent work items.  What makes them independent is kept by the forward: the per-pixel blend state (T, C, D) at every
// kSeg-th list position (a "snapshot", 20 bytes per pixel and boundary) - the backward's replay of positions [j kSeg, (j+1) kSeg)
// starts from the transmittance in front of position (j+1) kSeg and from the suffix colour (C_final - C_prefix) / T, both taken
// from the snapshots instead of from the replay of everything behind.  No running value of the replay feeds a discrete
// decision, so the segments take the decisions of the whole-list replay; sums differ by rounding only.
#ifndef T4D_SEG
#define T4D_SEG 128
#endif
constexpr int kSeg = T4D_SEG;        // list positions per backward segment (= kBwdBatch: one staged batch per segment)
constexpr int kSegMaxTiles = 8192;   // launches of at most this many tiles (V * T) run the segmented backward
constexpr int kSnapFloats = 5;       // T, C0, C1, C2, D per pixel and boundary

thread_local char g_err[512] = "";

// optional per-kernel timing with HIP events (t4d_profile_begin/end); used by bench.py for the roofline object
enum KernelId { K_PREPROCESS = 0, K_SCAN_TILES, K_SCATTER, K_SORT_TILES, K_RENDER_FWD, K_RENDER_BWD, K_PREPROCESS_BWD, K_COUNT };
const char *const kKernelNames[K_COUNT] = { "k_preprocess", "k_scan_tiles", "k_scatter", "k_sort_tiles", "k_render_fwd",
                                            "k_render_bwd", "k_preprocess_bwd" };
struct ProfRec { int id; hipEvent_t a, b; };
bool g_prof_on = false;
std::vector<ProfRec> g_prof;

// ---------------------------------------------------------------------------------------------------------
// state / scratch layout
// ---------------------------------------------------------------------------------------------------------
struct Layout {
    size_t status, view_total, view_cursor, tile_count, bucket_fill, slot_tab, zero_end;
    size_t snap;
    size_t tile_off, chunk_sum, order, items, xy, depth, conic_opacity, rgb, clamped, pair_off, pair_rank, keys, sort_tmp, final_T, n_contrib, total;
};

struct DevStatus {            // first bytes of the state buffer
    uint32_t overflow;
    uint32_t max_pairs;
    unsigned long long total_pairs;       // ... the 16 bytes T4D_FLAG_ASYNC_STATUS copies out end here
    uint32_t max_tile_pairs;              // longest tile list of the call (reported as T4DStatus.max_tile_pairs)
    uint32_t grid_sync;                   // arrival counter of k_front_small's one grid-wide barrier
};

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// segmented backward (see kSeg): decided by the problem's dimensions alone, so that t4d_state_bytes, the forward and the
// backward of a call agree without talking to each other
inline bool seg_capable(const T4DProblem &p)
{
    const long long T = (long long)((p.W + T4D_TILE_X - 1) / T4D_TILE_X) * (long long)((p.H + T4D_TILE_Y - 1) / T4D_TILE_Y);
    return (long long)p.n_views * T <= kSegMaxTiles;
}
// Segment slots of a view.  Tile t (arena offset off, n pairs) owns the slots floor(off / kSeg) + t ... + ceil(n / kSeg) - 1:
// disjoint from tile to tile ((off + n) / kSeg - off / kSeg >= floor(n / kSeg)) without a prefix sum over the tiles.
inline size_t seg_slots_per_view(const T4DProblem &p, size_t T) { return (size_t)p.pair_capacity / kSeg + T + 1; }

Layout make_layout(const T4DProblem &p)
{
    Layout L;
    const size_t V = (size_t)p.n_views, P = (size_t)p.P;
    const size_t T = (size_t)((p.W + T4D_TILE_X - 1) / T4D_TILE_X) * (size_t)((p.H + T4D_TILE_Y - 1) / T4D_TILE_Y);
    const size_t HW = (size_t)p.H * p.W;
    const size_t cap = (size_t)p.pair_capacity;
    size_t o = 0;
    L.status = o;        o = align_up(o + sizeof(DevStatus));
    L.view_total = o;    o = align_up(o + V * 4);
    L.view_cursor = o;   o = align_up(o + V * kCursorSegs * 4);
    L.tile_count = o;    o = align_up(o + V * T * 4);
    L.bucket_fill = o;   o = align_up(o + kBuckets * 4);
    const size_t slots = seg_capable(p) ? V * seg_slots_per_view(p, T) : 0;
    L.slot_tab = o;      o = align_up(o + slots * 16);           // (zeroed with the counters: an all-zero entry is "no segment")
    L.zero_end = o;
    L.snap = o;          o = align_up(o + slots * kSnapFloats * kBlock * 4);
    L.tile_off = o;      o = align_up(o + V * T * 4);
    L.chunk_sum = o;     o = align_up(o + V * ((T + kScanChunk - 1) / kScanChunk) * 4);
    L.order = o;         o = align_up(o + (size_t)kBuckets * V * T * 4);
    L.items = o;         o = align_up(o + V * T * 16);
    L.xy = o;            o = align_up(o + V * P * 8);
    L.depth = o;         o = align_up(o + V * P * 4);
    L.conic_opacity = o; o = align_up(o + V * P * 16);
    L.rgb = o;           o = align_up(o + (p.sh_coeffs > 0 ? V * P * 12 : 0));
    L.clamped = o;       o = align_up(o + (p.sh_coeffs > 0 ? V * P : 0));
    L.pair_off = o;      o = align_up(o + V * P * 4);
    L.pair_rank = o;     o = align_up(o + V * cap * 4);
    L.keys = o;          o = align_up(o + V * cap * 8);
    L.sort_tmp = o;      o = align_up(o + V * cap * 8);
    L.final_T = o;       o = align_up(o + V * HW * 4);
    L.n_contrib = o;     o = align_up(o + V * HW * 4);
    L.total = o;
    return L;
}

// ---------------------------------------------------------------------------------------------------------
// kernel parameter block (passed by value)
// ---------------------------------------------------------------------------------------------------------
struct KP {
    int V, P, H, W, gx, gy, T, deg, M;
    float scale_modifier;
    uint32_t cap;
    uint32_t nseg, seg_cap;          // the pair-slot arena of a view is split into nseg segments of seg_cap slots, one cursor each
    const float *views, *means3D, *opacities, *scales, *rotations, *cov3D_precomp, *colors_precomp, *shs;
    // state
    DevStatus *status;
    uint32_t *view_total, *view_cursor, *tile_count, *bucket_fill, *tile_off, *chunk_sum, *order, *pair_off, *pair_rank;
    int n_chunks;                    // scan chunks per view = ceil(T / kScanChunk)
    int long_bins_elsewhere;         // 1: k_sort_long runs behind k_sort_tiles and takes the bins longer than kSortLdsCap
    uint4 *items;
    float2 *xy;
    float *depth;
    float4 *conic_opacity;
    float *rgb;
    uint8_t *clamped;
    unsigned long long *keys, *sort_tmp;       // sort_tmp: ping-pong arena for bins longer than the LDS sort buffer
    float *cut_r2;                             // [V,cap] squared cut-off radius of every sorted pair (cutoff_radius2): written by the forward's staging, read by the backward's (lives in the pair_rank arena, which is dead once k_scatter has run)
    float *final_T;
    uint32_t *n_contrib;
    // forward outputs
    float *out_color, *out_depth, *out_alpha;
    int32_t *radii;
    // backward
    const float *dL_dcolor, *dL_ddepth, *dL_dalpha;
    float *grad_pair;
    float *dL_dmeans3D, *dL_dmeans2D, *dL_dcolors, *dL_dshs, *dL_dopacities, *dL_dscales, *dL_drotations, *dL_dcov3D;
    float *tile_dot;         // [V,T,4] per-wave <outputs, cotangents> of every tile (scratch) or nullptr when the caller did not ask
    float *cotangent_dot;    // [V]
    uint32_t tile_blocks;    // k_render_bwd / k_render_fwd: workgroups that walk the tile list (the backward's others do the empty tiles' dots)
    uint32_t fill_blocks;    // k_render_fwd: workgroups that write the EMPTY tiles' pixels, one per (view, row of tiles)
    uint32_t fill_vec;       // ... with 16-byte stores (W % 4 == 0 and 16-byte aligned output planes)
    // segmented backward of small launches (kSeg): slot table (id, arena offset, n, segment | 1 << 31) and forward snapshots
    uint4 *slot_tab;
    float *snap;             // [V * slots_per_view][kSnapFloats][256]
    uint32_t slots_per_view; // 0: this launch is not segmented
    uint32_t fused_sort;     // k_render_fwd<LAT = true> sorts its tile's bin itself (no k_sort_tiles launch)
    unsigned long long *host_status;  // T4D_FLAG_ASYNC_STATUS on a one-view launch: the caller's pinned 16 bytes, written by the kernel itself
};

// ---------------------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float ndc2pix(float v, int S)
{
#pragma clang fp contract(off)      // (v + 1) * S - 1 must not become an fma: S is not a power of two for 512x375 / 4096x3008 images
    return ((v + 1.0f) * S - 1.0f) * 0.5f;
}

__device__ __forceinline__ void tile_rect(float px, float py, int r, int gx, int gy, int &x0, int &y0, int &x1, int &y1)
{
    x0 = min(gx, max(0, (int)((px - r) / T4D_TILE_X)));
    y0 = min(gy, max(0, (int)((py - r) / T4D_TILE_Y)));
    x1 = min(gx, max(0, (int)((px + r + T4D_TILE_X - 1) / T4D_TILE_X)));
    y1 = min(gy, max(0, (int)((py + r + T4D_TILE_Y - 1) / T4D_TILE_Y)));
}

// A view's camera record, read through the CONSTANT address space: the record is the same for every lane of a workgroup, and
// only loads from memory the compiler knows to be read-only become scalar loads (s_load_dwordx16 into SGPRs).  Through a plain
// pointer the 40 floats came as per-lane vector loads: 32 vector registers for two matrices every lane holds identically, and
// one more level in the per-Gaussian kernels' chains of dependent loads.
typedef const __attribute__((address_space(4))) float *const_float_p;
struct ViewRecord {
    float view[16], proj[16], campos[3], bg[3], tanx, tany;
};
__device__ __forceinline__ ViewRecord load_view_record(const float *views, const int v)
{
    const_float_p p = (const_float_p)(views + (size_t)v * T4D_VIEW_FLOATS);
    ViewRecord r;
#pragma unroll
    for (int i = 0; i < 16; i++) { r.view[i] = p[i]; r.proj[i] = p[16 + i]; }
#pragma unroll
    for (int i = 0; i < 3; i++) { r.campos[i] = p[32 + i]; r.bg[i] = p[35 + i]; }
    r.tanx = p[38]; r.tany = p[39];
    return r;
}

// wave64 inclusive prefix sum (uint32)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

// DPP move helper: returns src permuted by CTRL; lanes/rows disabled by the masks read 0.
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ float dpp0(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, false));
}

// wave64 sum; the total is valid in lane 63 (CDNA row/bcast DPP: 6 v_add_f32_dpp, no LDS traffic)
__device__ __forceinline__ float wave_sum_to_lane63(float v)
{
    v += dpp0<0xB1>(v);          // quad_perm [1,0,3,2]
    v += dpp0<0x4E>(v);          // quad_perm [2,3,0,1]
    v += dpp0<0x141>(v);         // row_half_mirror
    v += dpp0<0x140>(v);         // row_mirror         -> every lane holds its 16-lane row sum
    v += dpp0<0x142, 0xA>(v);    // row_bcast15 into rows 1,3
    v += dpp0<0x143, 0xC>(v);    // row_bcast31 into rows 2,3 -> lane 63 = wave sum
    return v;
}

__device__ __forceinline__ void quat_rot(const float4 q, float R[9])
{
#pragma clang fp contract(off)
    const float r = q.x, x = q.y, y = q.z, z = q.w;
    R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - r * z);       R[2] = 2.f * (x * z + r * y);
    R[3] = 2.f * (x * y + r * z);       R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - r * x);
    R[6] = 2.f * (x * z - r * y);       R[7] = 2.f * (y * z + r * x);       R[8] = 1.f - 2.f * (x * x + y * y);
}

__device__ __forceinline__ void cov3d_from_scale_rot(const float *scale, float mod, const float4 q, float cov[6])
{
#pragma clang fp contract(off)
    float R[9], M[9];
    quat_rot(q, R);
    const float s[3] = { mod * scale[0], mod * scale[1], mod * scale[2] };
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int k = 0; k < 3; k++) M[i * 3 + k] = R[i * 3 + k] * s[k];
    cov[0] = M[0] * M[0] + M[1] * M[1] + M[2] * M[2];
    cov[1] = M[0] * M[3] + M[1] * M[4] + M[2] * M[5];
    cov[2] = M[0] * M[6] + M[1] * M[7] + M[2] * M[8];
    cov[3] = M[3] * M[3] + M[4] * M[4] + M[5] * M[5];
    cov[4] = M[3] * M[6] + M[4] * M[7] + M[5] * M[8];
    cov[5] = M[6] * M[6] + M[7] * M[7] + M[8] * M[8];
}

// rows T0,T1 of T = J*W of the EWA projection; t = clamped view-space point; in-range flags of the clamp
__device__ __forceinline__ void ewa_rows(const float *mean, const float *view, float fx, float fy, float tanx, float tany,
                                         float T0[3], float T1[3], float t[3], bool &inx, bool &iny)
{
#pragma clang fp contract(off)
    t[0] = view[0] * mean[0] + view[4] * mean[1] + view[8] * mean[2] + view[12];
    t[1] = view[1] * mean[0] + view[5] * mean[1] + view[9] * mean[2] + view[13];
    t[2] = view[2] * mean[0] + view[6] * mean[1] + view[10] * mean[2] + view[14];
    const float limx = T4D_FRUSTUM_CLAMP * tanx, limy = T4D_FRUSTUM_CLAMP * tany;
    const float txtz = t[0] / t[2], tytz = t[1] / t[2];
    inx = !(txtz < -limx || txtz > limx);
    iny = !(tytz < -limy || tytz > limy);
    t[0] = fminf(limx, fmaxf(-limx, txtz)) * t[2];
    t[1] = fminf(limy, fmaxf(-limy, tytz)) * t[2];
    const float J00 = fx / t[2], J02 = -(fx * t[0]) / (t[2] * t[2]);
    const float J11 = fy / t[2], J12 = -(fy * t[1]) / (t[2] * t[2]);
#pragma unroll
    for (int j = 0; j < 3; j++) {
        T0[j] = J00 * view[j * 4 + 0] + J02 * view[j * 4 + 2];
        T1[j] = J11 * view[j * 4 + 1] + J12 * view[j * 4 + 2];
    }
}

__device__ __forceinline__ void sym3_mul(const float c[6], const float v[3], float o[3])
{
#pragma clang fp contract(off)
    o[0] = c[0] * v[0] + c[1] * v[1] + c[2] * v[2];
    o[1] = c[1] * v[0] + c[3] * v[1] + c[4] * v[2];
    o[2] = c[2] * v[0] + c[4] * v[1] + c[5] * v[2];
}

__device__ __forceinline__ void sh_basis(int deg, const float d[3], float b[16])
{
#pragma clang fp contract(off)
    const float x = d[0], y = d[1], z = d[2];
    b[0] = T4D_SH_C0;
    if (deg > 0) {
        b[1] = -T4D_SH_C1 * y; b[2] = T4D_SH_C1 * z; b[3] = -T4D_SH_C1 * x;
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            b[4] = T4D_SH_C2_0 * xy; b[5] = T4D_SH_C2_1 * yz; b[6] = T4D_SH_C2_2 * (2.f * zz - xx - yy);
            b[7] = T4D_SH_C2_3 * xz; b[8] = T4D_SH_C2_4 * (xx - yy);
            if (deg > 2) {
                b[9] = T4D_SH_C3_0 * y * (3.f * xx - yy);
                b[10] = T4D_SH_C3_1 * xy * z;
                b[11] = T4D_SH_C3_2 * y * (4.f * zz - xx - yy);
                b[12] = T4D_SH_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy);
                b[13] = T4D_SH_C3_4 * x * (4.f * zz - xx - yy);
                b[14] = T4D_SH_C3_5 * z * (xx - yy);
                b[15] = T4D_SH_C3_6 * x * (xx - 3.f * yy);
            }
        }
    }
}

__device__ __forceinline__ void sh_basis_grad(int deg, const float d[3], float bx[16], float by[16], float bz[16])
{
    const float x = d[0], y = d[1], z = d[2];
#pragma unroll
    for (int k = 0; k < 16; k++) bx[k] = by[k] = bz[k] = 0.f;
    if (deg > 0) {
        by[1] = -T4D_SH_C1; bz[2] = T4D_SH_C1; bx[3] = -T4D_SH_C1;
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z;
            bx[4] = T4D_SH_C2_0 * y; by[4] = T4D_SH_C2_0 * x;
            by[5] = T4D_SH_C2_1 * z; bz[5] = T4D_SH_C2_1 * y;
            bx[6] = T4D_SH_C2_2 * -2.f * x; by[6] = T4D_SH_C2_2 * -2.f * y; bz[6] = T4D_SH_C2_2 * 4.f * z;
            bx[7] = T4D_SH_C2_3 * z; bz[7] = T4D_SH_C2_3 * x;
            bx[8] = T4D_SH_C2_4 * 2.f * x; by[8] = T4D_SH_C2_4 * -2.f * y;
            if (deg > 2) {
                bx[9] = T4D_SH_C3_0 * 6.f * x * y;             by[9] = T4D_SH_C3_0 * (3.f * xx - 3.f * yy);
                bx[10] = T4D_SH_C3_1 * y * z;                  by[10] = T4D_SH_C3_1 * x * z;   bz[10] = T4D_SH_C3_1 * x * y;
                bx[11] = T4D_SH_C3_2 * -2.f * x * y;           by[11] = T4D_SH_C3_2 * (4.f * zz - xx - 3.f * yy);
                bz[11] = T4D_SH_C3_2 * 8.f * y * z;
                bx[12] = T4D_SH_C3_3 * -6.f * x * z;           by[12] = T4D_SH_C3_3 * -6.f * y * z;
                bz[12] = T4D_SH_C3_3 * (6.f * zz - 3.f * xx - 3.f * yy);
                bx[13] = T4D_SH_C3_4 * (4.f * zz - 3.f * xx - yy); by[13] = T4D_SH_C3_4 * -2.f * x * y;
                bz[13] = T4D_SH_C3_4 * 8.f * x * z;
                bx[14] = T4D_SH_C3_5 * 2.f * x * z;            by[14] = T4D_SH_C3_5 * -2.f * y * z;
                bz[14] = T4D_SH_C3_5 * (xx - yy);
                bx[15] = T4D_SH_C3_6 * (3.f * xx - 3.f * yy);  by[15] = T4D_SH_C3_6 * -6.f * x * y;
            }
        }
    }
}

// Launch index of the per-Gaussian kernels -> (block of 256 Gaussians, view).  The V workgroups of one block read the same
// parameter rows (at config 4: 48 KiB of SH coefficients).  Workgroup b runs on XCD b % 8 and every XCD has its own L2, so the
// index is decoded such that ALL V workgroups of a block land on the SAME XCD, back to back: blocks go in groups of eight
// (one per XCD), a group takes 8 V consecutive indices, view-major.  The rows then leave HBM once, not once per XCD (the
// view-fastest order of round 2) or once per view (block-fastest).  T4D_GB_ORDER: bit 0 k_preprocess, bit 1
// k_preprocess_bwd, bit 2 k_sh_bwd take this order (experiments; default: the two backward kernels - k_preprocess is a
// third SLOWER with it, 211 -> 288 us at config 4, although its fetch traffic falls).
#ifndef T4D_GB_ORDER
#define T4D_GB_ORDER 6
#endif
__device__ __forceinline__ bool block_and_view(const uint32_t b, const uint32_t V, const uint32_t nblocks, uint32_t &gb, uint32_t &v)
{
    const uint32_t per = 8u * V, grp = b / per, rem = b - grp * per;
    v = rem >> 3;
    gb = grp * 8u + (rem & 7u);
    return gb < nblocks;
}
__host__ __device__ inline unsigned gaussian_grid(const int P, const int V) { return (unsigned)((((P + kBlock - 1) / kBlock + 7) / 8) * 8 * V); }

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

    uint32_t tiles = 0;
    int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    if (g < kp.P) {
        const float mean[3] = { kp.means3D[3 * (size_t)g], kp.means3D[3 * (size_t)g + 1], kp.means3D[3 * (size_t)g + 2] };
        // (covariance parameters and opacity are requested together with the mean, not behind the near-plane test)
        float cov3_in[6] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f }, sc[3] = { 0.f, 0.f, 0.f };
        float4 q = make_float4(1.f, 0.f, 0.f, 0.f);
        if (kp.cov3D_precomp) {
#pragma unroll
            for (int k = 0; k < 6; k++) cov3_in[k] = kp.cov3D_precomp[6 * (size_t)g + k];
        } else {
            sc[0] = kp.scales[3 * (size_t)g]; sc[1] = kp.scales[3 * (size_t)g + 1]; sc[2] = kp.scales[3 * (size_t)g + 2];
            q = reinterpret_cast<const float4 *>(kp.rotations)[g];
        }
        const float opacity = kp.opacities[g];
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
                    if (kp.shs && T4D_ABL != 8) {
                        float d[3] = { mean[0] - vrec.campos[0], mean[1] - vrec.campos[1], mean[2] - vrec.campos[2] };
                        const float len = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
                        d[0] /= len; d[1] /= len; d[2] /= len;
                        float bas[16];
                        sh_basis(kp.deg, d, bas);
                        const int K = (kp.deg + 1) * (kp.deg + 1);
                        const float *sh = kp.shs + (size_t)g * kp.M * 3;
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

#if T4D_ABL == 6
    return;
#endif
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
#if T4D_ABL == 7
    return;
#endif
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
#if T4D_GB_ORDER & 1
    if (!block_and_view(blockIdx.x, (uint32_t)kp.V, nblocks, gb, vb)) return;          // padding of the last group of eight
#else
    vb = blockIdx.x / nblocks; gb = blockIdx.x - vb * nblocks;
    if (vb >= (uint32_t)kp.V) return;
#endif
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
__device__ __forceinline__ void write_segment_slots(const KP &kp, const uint32_t id, const uint32_t off, const uint32_t n)
{
    if (kp.slots_per_view == 0u || n == 0u) return;
    const uint32_t nseg = (n + kSeg - 1) / kSeg;
    uint4 *tab = kp.slot_tab + (size_t)(id >> 20) * kp.slots_per_view + off / kSeg + (id & 0xfffffu);
    for (uint32_t j = 0; j < nseg; j++) tab[j] = make_uint4(id, off, n, j | 0x80000000u);
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
        write_segment_slots(kp, id, off, n);
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
                write_segment_slots(kp, (uint32_t)t, off[j], n);
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
// through memory less per forward: 9.9 + 7.7 us -> see DESIGN.md section 5.
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
                write_segment_slots(kp, (uint32_t)t, off[j], n);
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

// ---------------------------------------------------------------------------------------------------------
// A.2 per-tile sort by (depth bits, Gaussian index)
// ---------------------------------------------------------------------------------------------------------
// Sort the 64 keys of a wave (one per lane) ascending, entirely in registers: bitonic network whose exchanges are DPP
// moves (xor 1, 2: quad_perm; xor 4: two bank-masked row shifts; xor 8: row_ror:8) or ds_bpermute (xor 16, 32).
template <int J>
__device__ __forceinline__ uint32_t lane_xor(const uint32_t v)
{
    if (J == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);
    if (J == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);
    if (J == 4) {
        const int t = __builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xf, 0x5, true);      // banks {0,2} read lane+4
        return (uint32_t)__builtin_amdgcn_update_dpp(t, (int)v, 0x114, 0xf, 0xA, true);  // banks {1,3} read lane-4
    }
    if (J == 8) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, true);
    return (uint32_t)__shfl_xor((int)v, J, 64);
}

template <int K, int J>
__device__ __forceinline__ void bitonic_step(unsigned long long &key, const int lane)
{
    const unsigned long long other = ((unsigned long long)lane_xor<J>((uint32_t)(key >> 32)) << 32) | lane_xor<J>((uint32_t)key);
    const bool keep_min = ((lane & J) == 0) == ((lane & K) == 0);      // K = 64: every lane sorts ascending
    key = ((other < key) == keep_min) ? other : key;
    if constexpr (J > 1) bitonic_step<K, J / 2>(key, lane);
}

__device__ __forceinline__ void wave_sort64(unsigned long long &key, const int lane)
{
    bitonic_step<2, 1>(key, lane);
    bitonic_step<4, 2>(key, lane);
    bitonic_step<8, 4>(key, lane);
    bitonic_step<16, 8>(key, lane);
    bitonic_step<32, 16>(key, lane);
    bitonic_step<64, 32>(key, lane);
}

// number of keys smaller than `key` in a sorted run of 64 (branch-free binary search, 7 LDS reads)
__device__ __forceinline__ uint32_t run_lower_bound(const unsigned long long *run, const unsigned long long key)
{
    uint32_t pos = 0;
#pragma unroll
    for (int st = 32; st > 0; st >>= 1)
        if (run[pos + st - 1] < key) pos += st;
    return pos + (run[pos] < key ? 1u : 0u);
}

// number of keys smaller than `key` among run[0..len) (sorted, global memory)
__device__ __forceinline__ uint32_t lower_bound_global(const unsigned long long *run, const uint32_t len, const uint32_t cap2,
                                                       const unsigned long long key)
{
    uint32_t pos = 0;
    for (uint32_t st = cap2 >> 1; st > 0; st >>= 1)                  // cap2 = power of two >= len
        if (pos + st <= len && run[pos + st - 1] < key) pos += st;
    return pos + ((pos < len && run[pos] < key) ? 1u : 0u);
}

// Sort n <= kSortLdsCap keys (global memory, in place) through the workgroup's LDS buffer: runs of 64 are sorted in
// registers, then at every level each key finds its slot in the merged pair of runs as (position in its own run) + (keys of
// the sibling run below it), log2(width)+1 dependent LDS reads; keys wait in registers between the read and the write phase.
// log2(n/64) levels with two barriers each (a compare-exchange network needs ~60 barriers at this size).
template <int BLOCK, int CAP>
__device__ __forceinline__ void sort_chunk_lds(unsigned long long *keys, const uint32_t n, unsigned long long *s_keys,
                                               const int tid, const int wave, const int lane)
{
    constexpr int kPer = CAP / BLOCK;
    const uint32_t runs = (n + 63u) >> 6, N = runs << 6;
    for (uint32_t r = (uint32_t)wave; r < runs; r += BLOCK / 64) {
        const uint32_t i = (r << 6) + (uint32_t)lane;
        unsigned long long k0 = i < n ? keys[i] : ~0ull;           // the last run is padded with +inf
        wave_sort64(k0, lane);
        s_keys[i] = k0;
    }
    __syncthreads();
    for (uint32_t w = 64; w < N; w <<= 1) {
        unsigned long long kk[kPer];
        uint32_t np[kPer];
#pragma unroll
        for (int e = 0; e < kPer; e++) {
            const uint32_t p = (uint32_t)tid + e * BLOCK;
            if (p < N) {
                kk[e] = s_keys[p];
                const uint32_t run = p / w, sbase = (run ^ 1u) * w;
                const uint32_t slen = sbase < N ? min(w, N - sbase) : 0u;
                const unsigned long long *sib = s_keys + sbase;
                uint32_t pos = 0;
                for (uint32_t st = w >> 1; st > 0; st >>= 1)
                    if (pos + st <= slen && sib[pos + st - 1] < kk[e]) pos += st;
                if (pos < slen && sib[pos] < kk[e]) pos++;
                np[e] = (run & ~1u) * w + (p & (w - 1u)) + pos;
            }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < kPer; e++)
            if ((uint32_t)tid + e * BLOCK < N) s_keys[np[e]] = kk[e];
        __syncthreads();
    }
    for (uint32_t i = tid; i < n; i += BLOCK) keys[i] = s_keys[i];
    __syncthreads();
}

// Sort a bin longer than the LDS buffer: chunks of CAP keys are sorted through the LDS, then merged level by level IN GLOBAL
// MEMORY by the same ranking step, ping-pong between the key arena and the scratch arena of the same size (the bin's keys
// stay in this XCD's L2).  Four independent binary searches per thread and step overlap their latencies.
template <int BLOCK, int CAP>
__device__ __forceinline__ void sort_bin_chunked(unsigned long long *keys, unsigned long long *tmp, const uint32_t n,
                                                 unsigned long long *s_keys, const int tid, const int wave, const int lane)
{
    for (uint32_t c = 0; c < n; c += CAP) sort_chunk_lds<BLOCK, CAP>(keys + c, min((uint32_t)CAP, n - c), s_keys, tid, wave, lane);
    unsigned long long *src = keys, *dst = tmp;
    for (uint32_t w = CAP; w < n; w <<= 1) {
        __threadfence_block();
        __syncthreads();                                   // the previous level's writes are visible to the workgroup
        for (uint32_t i0 = (uint32_t)tid * 4u; i0 < n; i0 += BLOCK * 4u) {
            unsigned long long kk[4];
            uint32_t slot[4];
#pragma unroll
            for (int e = 0; e < 4; e++) kk[e] = i0 + e < n ? src[i0 + e] : ~0ull;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const uint32_t i = i0 + e, run = i / w, sbase = (run ^ 1u) * w;
                const uint32_t slen = sbase < n ? min(w, n - sbase) : 0u;
                slot[e] = (run & ~1u) * w + (i & (w - 1u)) + lower_bound_global(src + sbase, slen, w, kk[e]);
            }
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (i0 + e < n) dst[slot[e]] = kk[e];
        }
        unsigned long long *t2 = src; src = dst; dst = t2;
    }
    if (src != keys) {
        __threadfence_block();
        __syncthreads();
        for (uint32_t i = tid; i < n; i += BLOCK) keys[i] = src[i];
    }
}

// One bin of n keys, sorted by the whole workgroup (BLOCK threads) through s_keys (kSortLdsCap keys).  KEEP: leave the sorted keys
// in s_keys[0, n) as well (n <= kSortLdsCap) - the latency build of k_render_fwd sorts its own tile's bin and stages from there.
template <bool KEEP, int BLOCK>
__device__ __forceinline__ void sort_one_bin(const KP &kp, const int v, const uint32_t off, const uint32_t n, unsigned long long *s_keys,
                                             const int tid, const int wave, const int lane)
{
    unsigned long long *keys = kp.keys + (size_t)v * kp.cap + off;
    if (n <= (uint32_t)kRankSortMax) {
        // Runs of 64 keys are sorted inside a wave's registers (no LDS, no barrier); a key's final position is its
        // position in its own run plus, for every other run, the number of keys smaller than it (keys are unique: the
        // Gaussian index is the low word).  One barrier per tile, 7 dependent LDS reads per (key, other run).
        constexpr int kWaves = BLOCK / 64;
        constexpr int kPer = kRankSortMax / BLOCK > 0 ? kRankSortMax / BLOCK : 1;      // keys per thread
        const uint32_t runs = (n + 63u) >> 6;
        unsigned long long mine[kPer];
        uint32_t ranks[kPer];
#pragma unroll
        for (int e = 0; e < kPer; e++) {
            const uint32_t i = (uint32_t)tid + e * BLOCK;      // run (wave + kWaves e), position lane
            if ((uint32_t)(wave + kWaves * e) < runs) {        // wave-uniform
                mine[e] = i < n ? keys[i] : ~0ull;             // the last run is padded with +inf
                wave_sort64(mine[e], lane);
                s_keys[i] = mine[e];
            }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < kPer; e++) {
            const uint32_t own = (uint32_t)(wave + kWaves * e);
            ranks[e] = 0xffffffffu;
            if (own < runs) {
                uint32_t rank = (uint32_t)lane;
#pragma unroll
                for (uint32_t r = 0; r < (uint32_t)(kRankSortMax / 64); r++)      // unrolled: the searches overlap
                    if (r < runs && r != own) rank += run_lower_bound(s_keys + ((r % kWaves) * 64u + (r / kWaves) * BLOCK), mine[e]);
                if (mine[e] != ~0ull) { keys[rank] = mine[e]; ranks[e] = rank; }
            }
        }
        if (KEEP) {
            __syncthreads();                                   // every search has read the runs: they may be overwritten
#pragma unroll
            for (int e = 0; e < kPer; e++)
                if (ranks[e] != 0xffffffffu) s_keys[ranks[e]] = mine[e];
        }
    } else if (n <= (uint32_t)kSortLdsCap) {
        sort_chunk_lds<BLOCK, kSortLdsCap>(keys, n, s_keys, tid, wave, lane);
    } else if (!kp.long_bins_elsewhere) {
        // only when the host said that no such bin exists (T4D_FLAG_NO_LONG_BINS) and one appeared nevertheless:
        // correct, but one workgroup per bin with 16 KiB of LDS - k_sort_long is the fast path
        sort_bin_chunked<BLOCK, kSortLdsCap>(keys, kp.sort_tmp + (size_t)v * kp.cap + off, n, s_keys, tid, wave, lane);
    }
}

// BLOCK = 256: the throughput build (a 24-view launch holds thousands of bins: four waves per bin keep every SIMD busy).
// BLOCK = 1024: small launches (at most kSegMaxTiles tiles), whose sort lasts as long as its LONGEST bin takes one workgroup:
// a lone wave issues an instruction every four cycles, so a bin of 1,286 keys took 28 us on four waves (six register sorts of
// ~1 us and 5 merge levels of ~3 us per wave: tools/micro/sort_bin.hip); sixteen waves share that work.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_sort_tiles(const KP kp)
{
    constexpr int kWaves = BLOCK / 64;
    __shared__ unsigned long long s_keys[kSortLdsCap];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // Work units.  The item list is ordered by length class (floor(log2 n), descending), and bucket_fill holds the size of every
    // class: bins of 64 keys and more are one unit per workgroup; bins of 2..63 keys fit one register-sorted run, need neither LDS
    // nor a barrier, and go one PER WAVE to a unit (a high-resolution pass has mostly such bins: config 4 averages 65 keys
    // per non-empty tile, and three of the four waves of a one-bin workgroup did nothing).
    constexpr int kClass63 = (kBuckets - 2) - 5, kClass1 = kBuckets - 2;       // classes of n in [32, 63] and of n == 1
    uint32_t n_big = 0, n_small = 0;
#pragma unroll
    for (int k = 0; k < kBuckets - 1; k++) {
        const uint32_t f = kp.bucket_fill[k];
        if (k < kClass63) n_big += f;
        else if (k < kClass1) n_small += f;
    }
    const uint32_t n_units = n_big + ((n_small + kWaves - 1u) / kWaves);
    for (uint32_t unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
        if (unit >= n_big) {
            const uint32_t item = n_big + (uint32_t)kWaves * (unit - n_big) + (uint32_t)wave;     // wave-uniform
            if (item < n_big + n_small) {
                const uint4 it = kp.items[item];
                const uint32_t n = it.z;
                unsigned long long *keys = kp.keys + (size_t)(it.x >> 20) * kp.cap + it.y;
                unsigned long long k0 = (uint32_t)lane < n ? keys[lane] : ~0ull;
                wave_sort64(k0, lane);
                if ((uint32_t)lane < n) keys[lane] = k0;
            }
            continue;
        }
        const uint4 it = kp.items[unit];
        sort_one_bin<false, BLOCK>(kp, (int)(it.x >> 20), it.y, it.z, s_keys, tid, wave, lane);
        __syncthreads();                                           // s_keys is reused by the next item
    }
}

// Bins longer than kSortLdsCap keys (dense passes: 191 of 11,544 non-empty bins at P = 1M, 4096x3008, the longest 15,693 keys)
// get a whole CU each: 1024 threads and 128 KiB of LDS sort up to kLongCap keys without touching memory in between (runs of
// 64 in registers, then log2(n/64) ranking merges in LDS); even longer bins fall back to LDS-sorted chunks merged in global
// memory.  Work items are ordered by length class, so the long bins come first and a workgroup stops at the first bin of a shorter class.
__global__ __launch_bounds__(kLongBlock) void k_sort_long(const KP kp)
{
    __shared__ unsigned long long s_keys[kLongCap];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t n_items = (uint32_t)(kp.V * kp.T);
    for (uint32_t item = blockIdx.x; item < n_items; item += gridDim.x) {
        const uint4 it = kp.items[item];
        const int v = (int)(it.x >> 20);
        const uint32_t off = it.y, n = it.z;
        // The list is ordered by length CLASS (floor(log2 n)) only: a bin of exactly kSortLdsCap keys (k_sort_tiles' share) can
        // sit in front of longer bins of the same class, so it is skipped; the first bin of a shorter class ends the loop.
        if (n < (uint32_t)kSortLdsCap) break;
        if (n == (uint32_t)kSortLdsCap) continue;
        unsigned long long *keys = kp.keys + (size_t)v * kp.cap + off;
        if (n <= (uint32_t)kLongCap) sort_chunk_lds<kLongBlock, kLongCap>(keys, n, s_keys, tid, wave, lane);
        else sort_bin_chunked<kLongBlock, kLongCap>(keys, kp.sort_tmp + (size_t)v * kp.cap + off, n, s_keys, tid, wave, lane);
        __syncthreads();
    }
}

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

// Counting build (-DT4D_COUNT, tools/count_lanes.py; never defined in the shipped library): what the render kernels' visit loops
// do, summed over a launch - [0..7] backward, [8..15] forward:
//   +0 non-empty tiles   +1 live wave-batches   +2 wave-steps (one step = four DPP rows x 16 pixels)   +3 row-visits (list entries)
//   +4 lanes that blend / contribute (of 64 per wave-step)
#ifdef T4D_COUNT
__device__ unsigned long long g_count[16];
#define T4D_COUNT_ADD(IDX_, VAL_) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_count[(IDX_)], (unsigned long long)(VAL_)); } while (0)
#else
#define T4D_COUNT_ADD(IDX_, VAL_) do { } while (0)
#endif

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
// FB: splats staged per batch.  SEG: the launch is small enough for the segmented backward (kSeg): visit lists are built and
// walked per kSeg list positions, and the blend state at every such boundary is kept for the backward (write_snapshot).
// PRUNE: finished sub-blocks walk empty lists (below).  A template parameter because its mere presence costs the 72-register
// build 2.5 % at config 2 (register allocation, not executed instructions: a run-time gate that is never true costs the same),
// where no list is long enough for it to matter: the host instantiates it for launches that may hold long lists.
template <bool LAT, int FB, bool SEG, bool PRUNE>
__global__ __launch_bounds__(kBlock) T4D_FWD_ATTR void k_render_fwd(const KP kp)
{
    constexpr int kU = LAT ? 8 : 4;                  // steps per group
    // splats staged per batch: the latency build has the LDS of a whole CU and lives as long as its longest tile - fewer batches
    constexpr int kFB = FB;
    constexpr int kNull = kFB;                 // staged slot that can never contribute (opacity 0)
    constexpr int kSub = SEG ? kSeg : kFB;           // list positions per visit-list round
    constexpr int kSubChunks = kSub / 64, kNSub = kFB / kSub;
    constexpr int kListStride = kSub + 8;      // u16 entries per row list (multiple of 4: 8-byte aligned rows)
    constexpr int kRec = 48;                         // bytes per staged splat: xy, cut-off r2 (12, +4 pad) | scaled conic + opacity | rgb + depth
    static_assert(kFB <= kBlock && kFB % 64 == 0 && kFB % kSub == 0, "one staging thread per slot (it clears the slot when the list is shorter)");
    __shared__ __attribute__((aligned(16))) unsigned char s_rec[(kFB + 1) * kRec];
    __shared__ __attribute__((aligned(8))) unsigned short s_list[4][4][kListStride];
    __shared__ uint32_t s_wave_done[4];
    // latency build: the workgroup sorts its own tile's bin first (one launch and one trip through memory less than
    // k_sort_tiles -> k_render_fwd) and stages from the sorted keys it still holds
    __shared__ unsigned long long s_sort[LAT ? kSortLdsCap : 1];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, row = lane >> 4;
    // fill workgroups are spread evenly over the launch: workgroup b is one iff floor(b F / total) steps up at b
    const uint32_t total_blocks = kp.tile_blocks + kp.fill_blocks;
    const uint32_t fills_before = (uint32_t)(((unsigned long long)blockIdx.x * kp.fill_blocks) / total_blocks);
    if ((uint32_t)(((unsigned long long)(blockIdx.x + 1u) * kp.fill_blocks) / total_blocks) != fills_before) {
        fill_empty_tile_row(kp, fills_before);
        return;
    }
    if (tid < kRec / 4) reinterpret_cast<float *>(s_rec + kNull * kRec)[tid] = 0.f;
    for (uint32_t item = blockIdx.x - fills_before; item < (uint32_t)(kp.V * kp.T); item += kp.tile_blocks) {
    const uint4 it = kp.items[item];
    if (it.w == 0u) break;                           // ordered by length: only empty tiles remain, and those are not ours
    const int v = (int)(it.x >> 20), t_ = (int)(it.x & 0xfffffu);
    const int ty = t_ / kp.gx, tx = t_ - ty * kp.gx;
    const uint32_t off = it.y, n = it.z;
    const unsigned long long *keys = kp.keys + (size_t)v * kp.cap + off;
    float *r2_out = kp.cut_r2 + (size_t)v * kp.cap + off;
    const float2 *xy = kp.xy + (size_t)v * kp.P;
    const float4 *co = kp.conic_opacity + (size_t)v * kp.P;
    const float *rgb = kp.shs ? kp.rgb + (size_t)v * kp.P * 3 : kp.colors_precomp;
    // segmented backward: this tile's snapshot slots (a tile of one segment keeps none: its replay starts at the list's end)
    float *snap = nullptr;
    if (SEG && n > (uint32_t)kSeg)
        snap = kp.snap + ((size_t)v * kp.slots_per_view + off / kSeg + (uint32_t)t_) * (kSnapFloats * kBlock) + tid;

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
        if (b == 0 && wave == 0) T4D_COUNT_ADD(8, 1);
        T4D_COUNT_ADD(9, 1);
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
        T4D_COUNT_ADD(11, cnts[0] + cnts[1] + cnts[2] + cnts[3]);
#if T4D_ABL == 5
        nsteps = 0;
#endif
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
#if T4D_ABL == 4
            if (alpha[0] + alpha[1] + alpha[2] + alpha[3] == 12345.f) C0 += 1.f;
            continue;
#endif
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
                T4D_COUNT_ADD(12, __builtin_popcountll(live & ~below));
            }
            T4D_COUNT_ADD(10, kU);
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
#define T4D_BWD_NW (LAT ? 2 : (SEG ? T4D_SEG_WAVES : (DA ? T4D_BWD_DA_WAVES : T4D_BWD_WAVES)))
#define T4D_BWD_ATTR __attribute__((amdgpu_waves_per_eu(LAT ? 1 : T4D_BWD_NW, T4D_BWD_NW)))
constexpr int kAcc = 10;                 // sums per (wave, staged splat) slab entry
constexpr int kEmptySpan = 64;           // tiles per spare workgroup of the empty-tile share of cotangent_dot
// LAT: the latency build (see k_render_fwd): one slab per DPP ROW instead of one per wave (82 KB of LDS: one workgroup per CU
// is all such a launch has anyway), so two rows holding the same splat in the same step never meet and the conflict
// detection and its branches disappear; the gradient arithmetic is predicated with selects instead of an exec-masked region,
// which lets the compiler interleave the four steps of a group.
// SEG: the segmented backward of small launches (kSeg): a work item is ONE segment of a tile list - workgroup b takes slot b of
// the slot table - and the replay starts from the forward's snapshot at the segment's far end instead of from the list's end.
template <bool DA, bool LAT, bool SEG>
__global__ __launch_bounds__(kBlock) T4D_BWD_ATTR void k_render_bwd(const KP kp)
{
    static_assert(!SEG || kSeg == kBwdBatch, "one staged batch per segment");
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
    if (SEG) {
        it = kp.slot_tab[blockIdx.x];                // one slot per workgroup; most slots hold no segment
        if (it.w == 0u) return;
    }
    for (int i = tid; i < kSlabs * (kBwdBatch + 1) * kAcc; i += kBlock) (&s_acc[0][0][0])[i] = 0.f;   // slabs are all-zero between batches
    for (uint32_t item = blockIdx.x; item < (SEG ? blockIdx.x + 1u : (uint32_t)(kp.V * kp.T)); item += kp.tile_blocks) {
    if (!SEG) it = kp.items[item];
    const int seg_j = SEG ? (int)(it.w & 0x7fffffffu) : 0;           // this item's segment: list positions [seg_j kSeg, (seg_j + 1) kSeg)
    const int v = (int)(it.x >> 20), t_ = (int)(it.x & 0xfffffu);
    const int ty = t_ / kp.gx, tx = t_ - ty * kp.gx;
    const uint32_t off = it.y, n = it.z;
    if (n == 0) break;                                             // ordered by length: only empty tiles remain
    const unsigned long long *keys = kp.keys + (size_t)v * kp.cap + off;
    const float *r2_in = kp.cut_r2 + (size_t)v * kp.cap + off;
    const float2 *xy = kp.xy + (size_t)v * kp.P;
    const float4 *co = kp.conic_opacity + (size_t)v * kp.P;
    const float *rgb = kp.shs ? kp.rgb + (size_t)v * kp.P * 3 : kp.colors_precomp;
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
            const float *sb = kp.snap + ((size_t)v * kp.slots_per_view + off / kSeg + (uint32_t)t_) * (kSnapFloats * kBlock) + tid;
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
            T4D_COUNT_ADD(1, 1); T4D_COUNT_ADD(2, (nsteps + 3) & ~3); T4D_COUNT_ADD(3, cnts[0] + cnts[1] + cnts[2] + cnts[3]);
            if (bi == nb - 1 && wave == 0) T4D_COUNT_ADD(0, 1);
            // LAT: lanes that keep no sum write (zeros plus whatever) into distinct floats of the null splat's row of their slab
            unsigned char *slab = LAT ? reinterpret_cast<unsigned char *>(my_slot >= 0 ? &s_acc[wave * 4 + row][0][my_slot]
                                                                                      : &s_acc[wave * 4 + row][kNull][(lane & 15) % kAcc])
                                      : reinterpret_cast<unsigned char *>(&s_acc[wave][0][0] + (my_slot >= 0 ? my_slot : 0));
            const uint32_t slab_and = (!LAT || my_slot >= 0) ? 0xffffffffu : 0u;       // slot-less lanes of the latency build stay on their dummy float
            // entry of the first staged splat this pixel did NOT see in the forward pass (entries are slot * kEnt)
            const int lc_rel = (int)min(last_contributor - min(last_contributor, lo), (uint32_t)kBwdBatch) * kEnt;
#if T4D_ABL == 3
            nsteps = 0;
#endif
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
#ifdef T4D_COUNT
                    { const unsigned long long cb_ = __ballot(contrib); T4D_COUNT_ADD(4, __builtin_popcountll(cb_)); }
#endif
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
#ifdef T4D_RCP_NEWTON     // experiment (tools/ab_build.sh newton -DT4D_RCP_NEWTON): 1 / (1 - alpha) to within half an ulp; see DESIGN.md section 2
                        const float inv0 = __builtin_amdgcn_rcpf(om);
                        const float inv = fmaf(fmaf(-om, inv0, 1.f), inv0, inv0);
#else
                        const float inv = __builtin_amdgcn_rcpf(om);
#endif
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

// ---------------------------------------------------------------------------------------------------------
// A.5 per-Gaussian backward: gather pair records, then the chain rule down to the inputs
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void preprocess_bwd(const KP &kp)
{
    // One launch index, decoded by block_and_view: the V workgroups that read the same 256 Gaussians run back to back on one XCD
    // (round 2's view-fastest order: config 4 516 -> 484 us; all views on ONE XCD: another 1 %).  The same order makes
    // k_preprocess SLOWER (210 -> 290 us, measured in rounds 2 and 3) and is not used there.  The spare workgroups of the
    // per-view dot sit behind all of them.
    const uint32_t nblocks = (uint32_t)(kp.P + kBlock - 1) / kBlock;
    const uint32_t n_pv = gaussian_grid(kp.P, kp.V);
    const bool spare = blockIdx.x >= n_pv;
    uint32_t gb = 0, vb = 0;
#if T4D_GB_ORDER & 2
    if (!spare && !block_and_view(blockIdx.x, (uint32_t)kp.V, nblocks, gb, vb)) return;
#else
    gb = blockIdx.x / (uint32_t)kp.V; vb = blockIdx.x - gb * (uint32_t)kp.V;
    if (!spare && gb >= nblocks) return;
#endif
    const int v = spare ? (int)(blockIdx.x - n_pv) : (int)vb;
    const int g = (int)gb * kBlock + threadIdx.x;
    if (spare) {
        // one spare workgroup per view: the view's <outputs, cotangents> = sum of its tiles' dots, in a fixed order
        __shared__ float s_w[4];
        const float4 *td = reinterpret_cast<const float4 *>(kp.tile_dot) + (size_t)v * kp.T;      // one float per wave of the tile
        const float *vb = kp.views + (size_t)v * T4D_VIEW_FLOATS + 35;
        const bool black = vb[0] == 0.f && vb[1] == 0.f && vb[2] == 0.f;        // then nobody wrote the empty tiles' entries
        const uint32_t *tc = kp.tile_count + (size_t)v * kp.T;
        float a = 0.f;
        for (int t = threadIdx.x; t < kp.T; t += kBlock) {
            if (tc[t] == 0u && black) continue;
            const float4 d4 = td[t];
            a += (d4.x + d4.y) + (d4.z + d4.w);
        }
        a = wave_sum_to_lane63(a);
        if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = a;
        __syncthreads();
        if (threadIdx.x == 0) kp.cotangent_dot[v] = kp.status->overflow != 0u ? 0.f : (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
        return;
    }
    if (g >= kp.P) return;
    const size_t vg = (size_t)v * kp.P + g;
    const ViewRecord vrec = load_view_record(kp.views, v);
    const float *view = vrec.view, *proj = vrec.proj;
    // Everything that depends on (view, Gaussian) alone is requested HERE, before any of it is used: as the kernel was written
    // (radius, then centre, then pair slot, then the records, then conic and mean, then scale and rotation) a thread went through
    // eight dependent round trips, and the kernel is made of those (a quarter of the vector ALUs busy).
    const uint32_t flag = kp.status->overflow;
    const int radius_in = kp.radii[vg];
    const float2 p2 = kp.xy[vg];
    const uint32_t base = kp.pair_off[vg];
    const float4 cq = kp.conic_opacity[vg];
    const float mean[3] = { kp.means3D[3 * (size_t)g], kp.means3D[3 * (size_t)g + 1], kp.means3D[3 * (size_t)g + 2] };
    float cov3_in[6] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
    float4 q = make_float4(1.f, 0.f, 0.f, 0.f);
    float sc[3] = { 0.f, 0.f, 0.f };
    if (kp.cov3D_precomp) {
#pragma unroll
        for (int k = 0; k < 6; k++) cov3_in[k] = kp.cov3D_precomp[6 * (size_t)g + k];
    } else {
        sc[0] = kp.scales[3 * (size_t)g]; sc[1] = kp.scales[3 * (size_t)g + 1]; sc[2] = kp.scales[3 * (size_t)g + 2];
        q = reinterpret_cast<const float4 *>(kp.rotations)[g];
    }
    // A forward whose pair arena overflowed (possible only without T4D_FLAG_CHECKED) left tile lists truncated and pair
    // records unwritten: its backward returns ZERO gradients for every view instead of sums over uninitialised scratch.
    const bool truncated = flag != 0u;
    const int radius = truncated ? 0 : radius_in;

    float gm[3] = { 0.f, 0.f, 0.f }, g2x = 0.f, g2y = 0.f, gop = 0.f;
    float grgb[3] = { 0.f, 0.f, 0.f }, gsc[3] = { 0.f, 0.f, 0.f }, gq[4] = { 0.f, 0.f, 0.f, 0.f };
    float gcov[6] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };

    if (radius > 0) {
        // ---- gather the partial gradients of this Gaussian's tiles ----
        int x0, y0, x1, y1;
        tile_rect(p2.x, p2.y, radius, kp.gx, kp.gy, x0, y0, x1, y1);
        const uint32_t npairs = (uint32_t)((x1 - x0) * (y1 - y0));
        const float2 *gp = reinterpret_cast<const float2 *>(kp.grad_pair) + (size_t)v * kp.cap * (kGP / 2);
        float S0 = 0.f, S1 = 0.f, S2 = 0.f, S3 = 0.f, S4 = 0.f, S5 = 0.f, gdep = 0.f;
        for (uint32_t k = 0; k < npairs; k++) {
            const uint32_t pr = base + k;
            if (pr >= kp.cap) break;
            const float2 *rec = gp + (size_t)pr * (kGP / 2);           // 40-byte records: 8-byte aligned
            const float2 a0 = rec[0], a1 = rec[1], a2 = rec[2], a3 = rec[3], a4 = rec[4];
            S0 += a0.x; S1 += a0.y; S2 += a1.x; S3 += a1.y;
            S4 += a2.x; S5 += a2.y; grgb[0] += a3.x; grgb[1] += a3.y;
            grgb[2] += a4.x; gdep += a4.y;
        }
        // per-splat constants applied once (see k_render_bwd): dL/dG = opacity * dL/dalpha, dG/dd = -G * conic * d
        gop = S0;
        g2x = -cq.w * (cq.x * S1 + cq.y * S2) * (0.5f * kp.W);
        g2y = -cq.w * (cq.z * S2 + cq.y * S1) * (0.5f * kp.H);
        const float X = -0.5f * cq.w * S3, Y = -cq.w * S4, Z = -0.5f * cq.w * S5;      // true d/d(conic A, B, C)

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
        const float denom = a * c - b * b;
        const float d2inv = 1.f / ((denom * denom) + T4D_CONIC_BWD_EPS);
        float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
        if (d2inv != 0.f) {
            dL_da = d2inv * (-c * c * X + b * c * Y + (denom - a * c) * Z);
            dL_dc = d2inv * (-a * a * Z + a * b * Y + (denom - a * c) * X);
            dL_db = d2inv * (2.f * b * c * X - (denom + 2.f * b * b) * Y + 2.f * a * b * Z);
            gcov[0] = T0[0] * T0[0] * dL_da + T0[0] * T1[0] * dL_db + T1[0] * T1[0] * dL_dc;
            gcov[3] = T0[1] * T0[1] * dL_da + T0[1] * T1[1] * dL_db + T1[1] * T1[1] * dL_dc;
            gcov[5] = T0[2] * T0[2] * dL_da + T0[2] * T1[2] * dL_db + T1[2] * T1[2] * dL_dc;
            gcov[1] = 2.f * T0[0] * T0[1] * dL_da + (T0[0] * T1[1] + T0[1] * T1[0]) * dL_db + 2.f * T1[0] * T1[1] * dL_dc;
            gcov[2] = 2.f * T0[0] * T0[2] * dL_da + (T0[0] * T1[2] + T0[2] * T1[0]) * dL_db + 2.f * T1[0] * T1[2] * dL_dc;
            gcov[4] = 2.f * T0[2] * T0[1] * dL_da + (T0[1] * T1[2] + T0[2] * T1[1]) * dL_db + 2.f * T1[1] * T1[2] * dL_dc;
        }
        float dJ00 = 0.f, dJ02 = 0.f, dJ11 = 0.f, dJ12 = 0.f;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const float dT0 = 2.f * v0[j] * dL_da + v1[j] * dL_db;
            const float dT1 = 2.f * v1[j] * dL_dc + v0[j] * dL_db;
            dJ00 += view[j * 4 + 0] * dT0; dJ02 += view[j * 4 + 2] * dT0;
            dJ11 += view[j * 4 + 1] * dT1; dJ12 += view[j * 4 + 2] * dT1;
        }
        const float tz = 1.f / t[2], tz2 = tz * tz, tz3 = tz2 * tz;
        const float dtx = (inx ? 1.f : 0.f) * -fx * tz2 * dJ02;
        const float dty = (iny ? 1.f : 0.f) * -fy * tz2 * dJ12;
        const float dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2.f * fx * t[0]) * tz3 * dJ02 + (2.f * fy * t[1]) * tz3 * dJ12;
        gm[0] = view[0] * dtx + view[1] * dty + view[2] * dtz;
        gm[1] = view[4] * dtx + view[5] * dty + view[6] * dtz;
        gm[2] = view[8] * dtx + view[9] * dty + view[10] * dtz;

        // screen position -> mean (perspective divide)
        const float hx = proj[0] * mean[0] + proj[4] * mean[1] + proj[8] * mean[2] + proj[12];
        const float hy = proj[1] * mean[0] + proj[5] * mean[1] + proj[9] * mean[2] + proj[13];
        const float hw = proj[3] * mean[0] + proj[7] * mean[1] + proj[11] * mean[2] + proj[15];
        const float mw = 1.0f / (hw + T4D_HOM_W_EPS);
        const float mul1 = hx * mw * mw, mul2 = hy * mw * mw;
        gm[0] += (proj[0] * mw - proj[3] * mul1) * g2x + (proj[1] * mw - proj[3] * mul2) * g2y;
        gm[1] += (proj[4] * mw - proj[7] * mul1) * g2x + (proj[5] * mw - proj[7] * mul2) * g2y;
        gm[2] += (proj[8] * mw - proj[11] * mul1) * g2x + (proj[9] * mw - proj[11] * mul2) * g2y;
        // view depth -> mean
        gm[0] += view[2] * gdep; gm[1] += view[6] * gdep; gm[2] += view[10] * gdep;

        // colour: with precomputed RGB the pair sums ARE dL/dcolour; with SH colours they go to k_sh_bwd through a scratch
        // array (kp.dL_dcolors points at it), which also adds the view-direction term to dL/dmeans3D
        // cov3D -> scale, rotation
        if (!kp.cov3D_precomp) {
            float R[9];
            quat_rot(q, R);
            const float s[3] = { kp.scale_modifier * sc[0], kp.scale_modifier * sc[1], kp.scale_modifier * sc[2] };
            const float Gs[9] = { gcov[0], 0.5f * gcov[1], 0.5f * gcov[2], 0.5f * gcov[1], gcov[3], 0.5f * gcov[4],
                                  0.5f * gcov[2], 0.5f * gcov[4], gcov[5] };
            float Mp[9], D[9];
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int k = 0; k < 3; k++) Mp[r * 3 + k] = R[r * 3 + k] * s[k];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                float dM[3];
#pragma unroll
                for (int r = 0; r < 3; r++)
                    dM[r] = 2.f * (Gs[r * 3] * Mp[k] + Gs[r * 3 + 1] * Mp[3 + k] + Gs[r * 3 + 2] * Mp[6 + k]);
                gsc[k] = kp.scale_modifier * (dM[0] * R[k] + dM[1] * R[3 + k] + dM[2] * R[6 + k]);
#pragma unroll
                for (int r = 0; r < 3; r++) D[r * 3 + k] = dM[r] * s[k];
            }
            const float r = q.x, x = q.y, y = q.z, z = q.w;
            gq[0] = 2.f * z * (D[3] - D[1]) + 2.f * y * (D[2] - D[6]) + 2.f * x * (D[7] - D[5]);
            gq[1] = 2.f * y * (D[1] + D[3]) + 2.f * z * (D[2] + D[6]) + 2.f * r * (D[7] - D[5]) - 4.f * x * (D[4] + D[8]);
            gq[2] = 2.f * x * (D[1] + D[3]) + 2.f * r * (D[2] - D[6]) + 2.f * z * (D[5] + D[7]) - 4.f * y * (D[0] + D[8]);
            gq[3] = 2.f * r * (D[3] - D[1]) + 2.f * x * (D[2] + D[6]) + 2.f * y * (D[5] + D[7]) - 4.f * z * (D[0] + D[4]);
        }
    }

    kp.dL_dmeans3D[vg * 3] = gm[0]; kp.dL_dmeans3D[vg * 3 + 1] = gm[1]; kp.dL_dmeans3D[vg * 3 + 2] = gm[2];
    kp.dL_dmeans2D[vg * 3] = g2x; kp.dL_dmeans2D[vg * 3 + 1] = g2y; kp.dL_dmeans2D[vg * 3 + 2] = 0.f;
    kp.dL_dopacities[vg] = gop;
    if (kp.dL_dcolors) { kp.dL_dcolors[vg * 3] = grgb[0]; kp.dL_dcolors[vg * 3 + 1] = grgb[1]; kp.dL_dcolors[vg * 3 + 2] = grgb[2]; }
    if (kp.cov3D_precomp) {
#pragma unroll
        for (int k = 0; k < 6; k++) kp.dL_dcov3D[vg * 6 + k] = gcov[k];
    } else {
        kp.dL_dscales[vg * 3] = gsc[0]; kp.dL_dscales[vg * 3 + 1] = gsc[1]; kp.dL_dscales[vg * 3 + 2] = gsc[2];
        reinterpret_cast<float4 *>(kp.dL_drotations)[vg] = make_float4(gq[0], gq[1], gq[2], gq[3]);
    }
}

#ifndef T4D_PBWD_WAVES
#define T4D_PBWD_WAVES 6
#endif
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(T4D_PBWD_WAVES, T4D_PBWD_WAVES))) void k_preprocess_bwd(const KP kp)
{
    preprocess_bwd(kp);
}

// SH colours (BASELINE config 4): dL/dshs and the view-direction term of dL/dmeans3D, AFTER k_preprocess_bwd.  Rounds 1-2 did
// this inside the per-Gaussian kernel: 48 coefficients AND 48 gradients per thread in registers took it to 137 registers
// (three waves per SIMD), both with 192-byte lane strides - 484 us at config 4, a quarter of the vector ALUs busy.  Here:
//   1. one thread per (view, Gaussian): direction, basis, masked dL/dcolour; the coefficients stream through (they are only
//      needed for the gradient of the view direction, sum_k grad(basis_k) * (sh_k . dL/dcolour), which goes to dL/dmeans3D);
//      basis and dL/dcolour go to LDS;
//   2. the workgroup writes dL/dshs[k][c] = basis_k * dL/dcolour_c of its 256 Gaussians as ONE contiguous 48 KiB stream,
//      16 bytes per lane - 553 MB per step at config 4, the bulk of this kernel's traffic.
// Launch index view-fastest, like k_preprocess_bwd: the V workgroups that read the same coefficient rows run back to back.
__global__ __launch_bounds__(kBlock) void k_sh_bwd(const KP kp)
{
    __shared__ float s_bas[kBlock][17];                  // basis (odd pitch: one row per lane without bank conflicts)
    __shared__ float s_gc[kBlock][4];                    // masked dL/dcolour (zero for an invisible Gaussian)
    const int tid = threadIdx.x;
    const uint32_t nblocks = (uint32_t)(kp.P + kBlock - 1) / kBlock;
    uint32_t pblock, vb;
#if T4D_GB_ORDER & 4
    if (!block_and_view(blockIdx.x, (uint32_t)kp.V, nblocks, pblock, vb)) return;
#else
    pblock = blockIdx.x / (uint32_t)kp.V; vb = blockIdx.x - pblock * (uint32_t)kp.V;
    if (pblock >= nblocks) return;
#endif
    const int v = (int)vb;
    const int g0 = (int)pblock * kBlock;
    const int n = min(kBlock, kp.P - g0);                // Gaussians of this workgroup
    const int M3 = kp.M * 3;
    // ---- 1. per Gaussian
    if (tid < n) {
        const int g = g0 + tid;
        const size_t vg = (size_t)v * kp.P + g;
        // a truncated forward (arena overflow without T4D_FLAG_CHECKED) returns zero gradients everywhere
        const bool vis = kp.status->overflow == 0u && kp.radii[vg] > 0;
        float gc[3] = { 0.f, 0.f, 0.f };
        float bas[16];
#pragma unroll
        for (int i = 0; i < 16; i++) bas[i] = 0.f;
        if (vis) {
            const float *vr = kp.views + (size_t)v * T4D_VIEW_FLOATS;
            const float d0[3] = { kp.means3D[3 * (size_t)g] - vr[32], kp.means3D[3 * (size_t)g + 1] - vr[33], kp.means3D[3 * (size_t)g + 2] - vr[34] };
            const float len = sqrtf(d0[0] * d0[0] + d0[1] * d0[1] + d0[2] * d0[2]);
            const float d[3] = { d0[0] / len, d0[1] / len, d0[2] / len };
            float bx[16], by[16], bz[16];
            sh_basis(kp.deg, d, bas);
            sh_basis_grad(kp.deg, d, bx, by, bz);
            const uint32_t cl = kp.clamped[vg];                                   // channels the forward clamped at zero carry no gradient
            const float *grgb = kp.dL_dcolors + vg * 3;                           // the pair sums, left here by k_preprocess_bwd
            gc[0] = (cl & 1u) ? 0.f : grgb[0]; gc[1] = (cl & 2u) ? 0.f : grgb[1]; gc[2] = (cl & 4u) ? 0.f : grgb[2];
            const int K = (kp.deg + 1) * (kp.deg + 1);
            float gd[3] = { 0.f, 0.f, 0.f };
            const float *sh = kp.shs + (size_t)g * M3;
            if ((kp.M & 3) == 0 && K == 16) {            // degree 3, 16-byte aligned rows: twelve 16-byte loads, consumed as they come
                const float4 *sh4 = reinterpret_cast<const float4 *>(sh);
                float c[48];
#pragma unroll
                for (int i = 0; i < 12; i++) { const float4 t4 = sh4[i]; c[4 * i] = t4.x; c[4 * i + 1] = t4.y; c[4 * i + 2] = t4.z; c[4 * i + 3] = t4.w; }
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const float t = c[3 * k] * gc[0] + c[3 * k + 1] * gc[1] + c[3 * k + 2] * gc[2];
                    gd[0] += bx[k] * t; gd[1] += by[k] * t; gd[2] += bz[k] * t;
                }
            } else {
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    if (k < K) {
                        const float t = sh[k * 3] * gc[0] + sh[k * 3 + 1] * gc[1] + sh[k * 3 + 2] * gc[2];
                        gd[0] += bx[k] * t; gd[1] += by[k] * t; gd[2] += bz[k] * t;
                    } else {
                        bas[k] = 0.f;
                    }
                }
            }
            const float dot = d[0] * gd[0] + d[1] * gd[1] + d[2] * gd[2];         // the direction was normalised: project its gradient
            float *gm = kp.dL_dmeans3D + vg * 3;
#pragma unroll
            for (int jj = 0; jj < 3; jj++) gm[jj] += (gd[jj] - d[jj] * dot) / len;
        }
#pragma unroll
        for (int i = 0; i < 16; i++) s_bas[tid][i] = bas[i];
        s_gc[tid][0] = gc[0]; s_gc[tid][1] = gc[1]; s_gc[tid][2] = gc[2];
    }
    __syncthreads();
    // ---- 2. dL/dshs, as one contiguous stream
    float *out = kp.dL_dshs + ((size_t)v * kp.P + g0) * M3;
    if (kp.M == 16) {
        float4 *out4 = reinterpret_cast<float4 *>(out);
        for (int i = tid; i < n * 12; i += kBlock) {
            const int slot = i / 12, e0 = (i % 12) * 4;
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; e++) o[e] = s_bas[slot][(e0 + e) / 3] * s_gc[slot][(e0 + e) % 3];
            out4[i] = make_float4(o[0], o[1], o[2], o[3]);
        }
    } else {
        // (coefficients beyond degree 3 - M > 16 - have no basis function: zero gradient)
        for (int i = tid; i < n * M3; i += kBlock) {
            const int slot = i / M3, e = i % M3, k = e / 3;
            out[i] = k < 16 ? s_bas[slot][k] * s_gc[slot][e % 3] : 0.f;
        }
    }
}

// The same for the case that matters (degree 3, M = 16: BASELINE config 4), built around how the coefficient rows travel.  Above,
// every lane fetches its own 192-byte row with twelve 16-byte loads: one load instruction touches 64 different cache lines, the
// rows of all resident waves (240 KB per CU) do not survive in the 32 KB L1 from one load to the next, and every view fetches
// them again.  Here a workgroup takes its 256 rows ONCE, as one contiguous 48 KiB stream (16 bytes per lane, consecutive lanes
// consecutive addresses), turns them through LDS into one row per lane held in registers, and then serves T4D_SHB_VIEWS views
// from them; the staging area is reused for the basis / dL/dcolour exchange of the write-out.
#ifndef T4D_SHB_VIEWS
#define T4D_SHB_VIEWS 8
#endif
__global__ __launch_bounds__(kBlock) void k_sh_bwd16(const KP kp)
{
    constexpr int kPitch = 52;                           // floats per staged row: 16-byte aligned, 13 (odd) 16-byte words -> no bank conflicts
    __shared__ __attribute__((aligned(16))) float s_raw[kBlock * kPitch];
    float (*s_bas)[17] = reinterpret_cast<float (*)[17]>(s_raw);                   // after the staging: basis (odd pitch) ...
    float (*s_gc)[4] = reinterpret_cast<float (*)[4]>(s_raw + kBlock * 17);        // ... and masked dL/dcolour
    static_assert(kBlock * 17 + kBlock * 4 <= kBlock * kPitch && (kBlock * 17) % 4 == 0, "the exchange arrays live inside the staging area");
    const int tid = threadIdx.x;
    const uint32_t nblocks = (uint32_t)(kp.P + kBlock - 1) / kBlock;
    const uint32_t ngroups = (uint32_t)(kp.V + T4D_SHB_VIEWS - 1) / T4D_SHB_VIEWS;
    uint32_t pblock, vgrp;
    if (!block_and_view(blockIdx.x, ngroups, nblocks, pblock, vgrp)) return;
    const int g0 = (int)pblock * kBlock;
    const int n = min(kBlock, kp.P - g0);                // Gaussians of this workgroup
    {
        const float4 *src = reinterpret_cast<const float4 *>(kp.shs + (size_t)g0 * 48);
        for (int i = tid; i < n * 12; i += kBlock) {
            const int r = i / 12, part = i - r * 12;
            *reinterpret_cast<float4 *>(s_raw + r * kPitch + part * 4) = src[i];
        }
    }
    __syncthreads();
    float c[48];
#pragma unroll
    for (int i = 0; i < 12; i++) {
        const float4 t4 = *reinterpret_cast<const float4 *>(s_raw + min(tid, n - 1) * kPitch + 4 * i);
        c[4 * i] = t4.x; c[4 * i + 1] = t4.y; c[4 * i + 2] = t4.z; c[4 * i + 3] = t4.w;
    }
    __syncthreads();                                     // the rows are in registers: the staging area is free
    const int g = g0 + min(tid, n - 1);
    const float mean[3] = { kp.means3D[3 * (size_t)g], kp.means3D[3 * (size_t)g + 1], kp.means3D[3 * (size_t)g + 2] };
    const bool truncated = kp.status->overflow != 0u;   // a truncated forward (arena overflow without T4D_FLAG_CHECKED): zero gradients
    const int v_end = min(kp.V, (int)(vgrp + 1u) * T4D_SHB_VIEWS);
    for (int v = (int)vgrp * T4D_SHB_VIEWS; v < v_end; v++) {
        // ---- 1. per Gaussian: direction, basis, masked dL/dcolour, the view-direction term of dL/dmeans3D
        if (tid < n) {
            const size_t vg = (size_t)v * kp.P + g;
            const bool vis = !truncated && kp.radii[vg] > 0;
            float gc[3] = { 0.f, 0.f, 0.f };
            float bas[16];
#pragma unroll
            for (int i = 0; i < 16; i++) bas[i] = 0.f;
            if (vis) {
                const float *vr = kp.views + (size_t)v * T4D_VIEW_FLOATS;
                const float d0[3] = { mean[0] - vr[32], mean[1] - vr[33], mean[2] - vr[34] };
                const float len = sqrtf(d0[0] * d0[0] + d0[1] * d0[1] + d0[2] * d0[2]);
                const float d[3] = { d0[0] / len, d0[1] / len, d0[2] / len };
                float bx[16], by[16], bz[16];
                sh_basis(3, d, bas);
                sh_basis_grad(3, d, bx, by, bz);
                const uint32_t cl = kp.clamped[vg];                               // channels the forward clamped at zero carry no gradient
                const float *grgb = kp.dL_dcolors + vg * 3;                       // the pair sums, left here by k_preprocess_bwd
                gc[0] = (cl & 1u) ? 0.f : grgb[0]; gc[1] = (cl & 2u) ? 0.f : grgb[1]; gc[2] = (cl & 4u) ? 0.f : grgb[2];
                float gd[3] = { 0.f, 0.f, 0.f };
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const float t = c[3 * k] * gc[0] + c[3 * k + 1] * gc[1] + c[3 * k + 2] * gc[2];
                    gd[0] += bx[k] * t; gd[1] += by[k] * t; gd[2] += bz[k] * t;
                }
                const float dot = d[0] * gd[0] + d[1] * gd[1] + d[2] * gd[2];     // the direction was normalised: project its gradient
                float *gm = kp.dL_dmeans3D + vg * 3;
#pragma unroll
                for (int jj = 0; jj < 3; jj++) gm[jj] += (gd[jj] - d[jj] * dot) / len;
            }
#pragma unroll
            for (int i = 0; i < 16; i++) s_bas[tid][i] = bas[i];
            s_gc[tid][0] = gc[0]; s_gc[tid][1] = gc[1]; s_gc[tid][2] = gc[2];
        }
        __syncthreads();
        // ---- 2. dL/dshs of this view, as one contiguous stream
        float4 *out4 = reinterpret_cast<float4 *>(kp.dL_dshs + ((size_t)v * kp.P + g0) * 48);
        for (int i = tid; i < n * 12; i += kBlock) {
            const int slot = i / 12, e0 = (i - slot * 12) * 4;
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; e++) o[e] = s_bas[slot][(e0 + e) / 3] * s_gc[slot][(e0 + e) % 3];
            out4[i] = make_float4(o[0], o[1], o[2], o[3]);
        }
        __syncthreads();                                 // the exchange arrays are rewritten by the next view
    }
}

// ---------------------------------------------------------------------------------------------------------
// per-view scalar <a, b> (e.g. the loss term sum(colour * dL/dcolour) each rank contributes to the loss gather):
// one pass over both images, deterministic two-level sum.
// ---------------------------------------------------------------------------------------------------------
constexpr int kDotBlocks = 64;

__global__ __launch_bounds__(kBlock) void k_view_dot_partial(const float *a, const float *b, size_t n, float *partial)
{
    __shared__ float s_w[4];
    const int v = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
    const float *pa = a + (size_t)v * n, *pb = b + (size_t)v * n;
    const size_t per = (((n + kDotBlocks - 1) / kDotBlocks) + 3) & ~(size_t)3;     // multiple of 4
    const size_t lo = min(n, (size_t)blk * per), hi = min(n, lo + per);
    float acc = 0.f;
    if ((n & 3) == 0 && (((uintptr_t)a | (uintptr_t)b) & 15) == 0) {               // 16-byte loads
        for (size_t i = lo + (size_t)tid * 4; i < hi; i += (size_t)kBlock * 4) {
            const float4 x = *reinterpret_cast<const float4 *>(pa + i), y = *reinterpret_cast<const float4 *>(pb + i);
            acc += (x.x * y.x + x.y * y.y) + (x.z * y.z + x.w * y.w);
        }
    } else {
        for (size_t k = lo + tid; k < hi; k += kBlock) acc += pa[k] * pb[k];
    }
    acc = wave_sum_to_lane63(acc);
    if ((tid & 63) == 63) s_w[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) partial[(size_t)v * kDotBlocks + blk] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}

__global__ __launch_bounds__(64) void k_view_dot_final(const float *partial, float *out)
{
    const int v = blockIdx.x;
    float x = partial[(size_t)v * kDotBlocks + threadIdx.x];
    x = wave_sum_to_lane63(x);
    if (threadIdx.x == 63) out[v] = x;
}

__global__ __launch_bounds__(kBlock) void k_mark_visible(int P, const float *means3D, const float *view, uint8_t *present)
{
#pragma clang fp contract(off)
    const int g = blockIdx.x * kBlock + threadIdx.x;
    if (g >= P) return;
    const float z = view[2] * means3D[3 * (size_t)g] + view[6] * means3D[3 * (size_t)g + 1] +
                    view[10] * means3D[3 * (size_t)g + 2] + view[14];
    present[g] = z > T4D_NEAR_CULL_Z ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
int fail(int code, const char *fmt, const char *a = "")
{
    snprintf(g_err, sizeof(g_err), fmt, a);
    return code;
}
}  // namespace

// shared with the other translation units of this library (hidden visibility: not part of the ABI)
int t4d_internal_fail(int code, const char *fmt, const char *a) { return fail(code, fmt, a); }

namespace {

#define T4D_HIP(call)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) return fail(T4D_ERR_HIP, #call ": %s", hipGetErrorString(e_));  \
    } while (0)

struct ProfScope {
    hipStream_t s; int id; hipEvent_t a = nullptr, b = nullptr; bool on;
    ProfScope(hipStream_t s_, int id_) : s(s_), id(id_), on(g_prof_on)
    {
        if (on) { (void)hipEventCreate(&a); (void)hipEventCreate(&b); (void)hipEventRecord(a, s); }
    }
    ~ProfScope()
    {
        if (on) { (void)hipEventRecord(b, s); g_prof.push_back({ id, a, b }); }
    }
};

#define T4D_LAUNCH_CHECK(name)                                                                 \
    do {                                                                                       \
        hipError_t e_ = hipGetLastError();                                                     \
        if (e_ != hipSuccess) return fail(T4D_ERR_HIP, name " launch: %s", hipGetErrorString(e_)); \
        if (debug) {                                                                           \
            e_ = hipStreamSynchronize(stream);                                                 \
            if (e_ != hipSuccess) return fail(T4D_ERR_HIP, name " exec: %s", hipGetErrorString(e_)); \
        }                                                                                      \
    } while (0)

int device_cus()
{
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;                                             // MI355X
        cus = n;
    }
    return cus;
}

// Grid of the per-tile kernels.  They are grid-stride loops over the length-ordered work items, so the grid size is a
// free choice.  Measured on MI355X (config 2): a RESIDENT grid (CUs x 4-6 workgroups; the switch is gone) loses 1.3x to
// the hardware dispatcher's dynamic balancing; one workgroup per tile pays ~25k workgroup launches of which two
// thirds only find an empty tile; V*T / div workgroups, each taking items b, b+G, b+2G, ... (one from every length
// class, heavy first), keeps the dynamic balancing and divides the launch overhead.  T4D_TILE_DIV overrides div.
int tile_grid(int n_tiles, int per_cu, int div)
{
    static int env_div = -1;
    if (env_div < 0) {
        const char *d = getenv("T4D_TILE_DIV");
        env_div = d ? atoi(d) : 0;
    }
    if (env_div > 0) return max(min(n_tiles, device_cus() * per_cu), (n_tiles + env_div - 1) / env_div);
    // ... up to kMaxGridPerCu workgroups per CU: beyond that (config 4: 393k tiles per launch) more workgroups only add prologues -
    // the best div measured there was 12-16 (24-33k workgroups: 5.13 -> 4.79 ms per step), at config 2 (24.6k tiles) it is 2
    constexpr int kMaxGridPerCu = 96;
    return max(min(n_tiles, device_cus() * per_cu), min((n_tiles + div - 1) / div, device_cus() * kMaxGridPerCu));
}

// A launch of at most this many tiles runs the LATENCY builds of the render kernels: with <= 4 workgroups per CU in total
// nothing queues behind anything, and a kernel lasts as long as its longest tile (T4D_LATENCY_TILES overrides; 0 = never).
bool latency_launch(int n_tiles)
{
    const char *e = getenv("T4D_LATENCY_TILES");        // read per call: the tests switch builds at run time
    return n_tiles <= (e ? atoi(e) : 4 * device_cus());
}
// The FORWARD's latency build (registers for instruction-level parallelism, its own tile's bin sorted in the workgroup, records
// of the next batch in flight during the walk) stays ahead of the throughput build for much larger launches than the
// backward's did (which needs 82 KiB of LDS per workgroup): T4D_FWD_LATENCY_TILES overrides, T4D_LATENCY_TILES too (tests).
bool latency_launch_fwd(int n_tiles, uint32_t flags)
{
    const char *e = getenv("T4D_FWD_LATENCY_TILES");
    if (!e) e = getenv("T4D_LATENCY_TILES");
    // Where the two builds cross depends on the scene: a launch that waits for one long list (the config-2 scene: 1,400 pairs in a
    // tile) is better off in the latency build up to ~6 views (4 views: 67 against 82 us), a launch of many short lists (Topo4D's
    // scene, longest 470) only up to 4 (5 views: 48 against 35 us).  The caller's T4D_FLAG_LONG_LISTS says which it is.
    return n_tiles <= (e ? atoi(e) : ((flags & T4D_FLAG_LONG_LISTS) ? 24 : 12) * device_cus());
}

int check_problem(const T4DProblem *p)
{
    if (!p) return fail(T4D_ERR_ARG, "null problem");
    if (p->abi_version != T4D_ABI_VERSION) return fail(T4D_ERR_ARG, "abi_version mismatch");
    if (p->n_views < 1 || p->P < 1 || p->H < 1 || p->W < 1) return fail(T4D_ERR_ARG, "n_views, P, H, W must be >= 1");
    if (p->n_views > 4095) return fail(T4D_ERR_ARG, "n_views must be <= 4095");
    if ((int64_t)((p->W + T4D_TILE_X - 1) / T4D_TILE_X) * ((p->H + T4D_TILE_Y - 1) / T4D_TILE_Y) > 0xfffff)
        return fail(T4D_ERR_ARG, "image too large: more than 2^20 tiles");
    if ((int64_t)p->n_views * ((p->W + T4D_TILE_X - 1) / T4D_TILE_X) * ((p->H + T4D_TILE_Y - 1) / T4D_TILE_Y) > (1LL << 30))
        return fail(T4D_ERR_ARG, "n_views * tiles exceeds 2^30 work items: split the batch");
    if (p->pair_capacity < 1 || p->pair_capacity > 0x7fffffffLL) return fail(T4D_ERR_ARG, "pair_capacity out of range");
    if (p->sh_coeffs < 0 || p->sh_degree < 0 || p->sh_degree > 3) return fail(T4D_ERR_ARG, "sh_degree must be 0..3");
    if (p->sh_coeffs > 0 && p->sh_coeffs < (p->sh_degree + 1) * (p->sh_degree + 1))
        return fail(T4D_ERR_ARG, "sh_coeffs smaller than (sh_degree+1)^2");
    return T4D_OK;
}

void fill_common(KP &kp, const T4DProblem &p, const Layout &L, char *st)
{
    kp.V = p.n_views; kp.P = p.P; kp.H = p.H; kp.W = p.W;
    kp.gx = (p.W + T4D_TILE_X - 1) / T4D_TILE_X;
    kp.gy = (p.H + T4D_TILE_Y - 1) / T4D_TILE_Y;
    kp.T = kp.gx * kp.gy;
    kp.deg = p.sh_degree; kp.M = p.sh_coeffs;
    kp.scale_modifier = p.scale_modifier;
    kp.cap = (uint32_t)p.pair_capacity;
    {
        const uint32_t n_wg = (uint32_t)((p.P + kBlock - 1) / kBlock);
        kp.nseg = 1;
        while (kp.nseg < (uint32_t)kCursorSegs && n_wg >= 16u * kp.nseg) kp.nseg <<= 1;       // >= 8 workgroups per cursor
        kp.seg_cap = kp.cap / kp.nseg;
    }
    kp.status = reinterpret_cast<DevStatus *>(st + L.status);
    kp.view_total = reinterpret_cast<uint32_t *>(st + L.view_total);
    kp.view_cursor = reinterpret_cast<uint32_t *>(st + L.view_cursor);
    kp.tile_count = reinterpret_cast<uint32_t *>(st + L.tile_count);
    kp.bucket_fill = reinterpret_cast<uint32_t *>(st + L.bucket_fill);
    kp.order = reinterpret_cast<uint32_t *>(st + L.order);
    kp.items = reinterpret_cast<uint4 *>(st + L.items);
    kp.pair_rank = reinterpret_cast<uint32_t *>(st + L.pair_rank);
    kp.cut_r2 = reinterpret_cast<float *>(st + L.pair_rank);
    kp.tile_off = reinterpret_cast<uint32_t *>(st + L.tile_off);
    kp.chunk_sum = reinterpret_cast<uint32_t *>(st + L.chunk_sum);
    kp.n_chunks = (kp.T + kScanChunk - 1) / kScanChunk;
    kp.long_bins_elsewhere = (p.flags & T4D_FLAG_NO_LONG_BINS) ? 0 : 1;
    kp.pair_off = reinterpret_cast<uint32_t *>(st + L.pair_off);
    kp.xy = reinterpret_cast<float2 *>(st + L.xy);
    kp.depth = reinterpret_cast<float *>(st + L.depth);
    kp.conic_opacity = reinterpret_cast<float4 *>(st + L.conic_opacity);
    kp.rgb = reinterpret_cast<float *>(st + L.rgb);
    kp.clamped = reinterpret_cast<uint8_t *>(st + L.clamped);
    kp.keys = reinterpret_cast<unsigned long long *>(st + L.keys);
    kp.sort_tmp = reinterpret_cast<unsigned long long *>(st + L.sort_tmp);
    kp.final_T = reinterpret_cast<float *>(st + L.final_T);
    kp.n_contrib = reinterpret_cast<uint32_t *>(st + L.n_contrib);
    kp.slot_tab = reinterpret_cast<uint4 *>(st + L.slot_tab);
    kp.snap = reinterpret_cast<float *>(st + L.snap);
    kp.slots_per_view = seg_capable(p) ? (uint32_t)seg_slots_per_view(p, (size_t)kp.T) : 0u;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------
T4D_EXPORT uint32_t t4d_abi_version(void) { return T4D_ABI_VERSION; }
T4D_EXPORT const char *t4d_last_error(void) { return g_err; }

T4D_EXPORT size_t t4d_state_bytes(const T4DProblem *prob)
{
    if (check_problem(prob) != T4D_OK) return 0;
    return make_layout(*prob).total;
}

size_t grad_pair_bytes(const T4DProblem &p) { return align_up((size_t)p.n_views * (size_t)p.pair_capacity * kGP * sizeof(float)); }
size_t tile_dot_bytes(const T4DProblem &p, size_t n_tiles) { return align_up((size_t)p.n_views * n_tiles * 4 * sizeof(float)); }

T4D_EXPORT size_t t4d_backward_scratch_bytes(const T4DProblem *prob)
{
    if (check_problem(prob) != T4D_OK) return 0;
    const size_t n_tiles = (size_t)((prob->W + T4D_TILE_X - 1) / T4D_TILE_X) * ((prob->H + T4D_TILE_Y - 1) / T4D_TILE_Y);
    // pair records | per-wave tile dots | (SH colours) dL/dcolour per (view, Gaussian), handed from k_preprocess_bwd to k_sh_bwd
    return grad_pair_bytes(*prob) + tile_dot_bytes(*prob, n_tiles) +
           (prob->sh_coeffs > 0 ? align_up((size_t)prob->n_views * (size_t)prob->P * 3 * sizeof(float)) : 0);
}

T4D_EXPORT int t4d_debug_state_layout(const T4DProblem *prob, int has_sh, uint64_t *offsets, int n)
{
    int rc = check_problem(prob);
    if (rc != T4D_OK) return rc;
    if (!offsets || n < T4D_DEBUG_LAYOUT_FIELDS) return fail(T4D_ERR_ARG, "offsets must hold T4D_DEBUG_LAYOUT_FIELDS entries");
    T4DProblem p = *prob;
    if (!has_sh) p.sh_coeffs = 0;
    const Layout L = make_layout(p);
    const size_t f[T4D_DEBUG_LAYOUT_FIELDS] = { L.status, L.view_total, L.view_cursor, L.tile_count, L.bucket_fill, L.tile_off,
                                                L.xy, L.depth, L.conic_opacity, L.rgb, L.clamped, L.pair_off, L.keys,
                                                L.final_T, L.n_contrib, L.total };
    for (int i = 0; i < T4D_DEBUG_LAYOUT_FIELDS; i++) offsets[i] = (uint64_t)f[i];
    return T4D_OK;
}

T4D_EXPORT int t4d_rasterize_forward(const T4DProblem *prob, const T4DForwardIO *io, T4DStatus *status, void *hip_stream)
{
    int rc = check_problem(prob);
    if (rc != T4D_OK) return rc;
    if (!io || !io->views || !io->means3D || !io->opacities || !io->out_color || !io->out_depth || !io->out_alpha ||
        !io->out_radii || !io->state)
        return fail(T4D_ERR_ARG, "null required pointer in T4DForwardIO");
    if ((io->shs == nullptr) == (io->colors_precomp == nullptr))
        return fail(T4D_ERR_ARG, "provide exactly one of shs / colors_precomp");
    if (io->cov3D_precomp ? (io->scales || io->rotations) : (!io->scales || !io->rotations))
        return fail(T4D_ERR_ARG, "provide exactly one of (scales, rotations) / cov3D_precomp");
    if (io->shs && prob->sh_coeffs < 1) return fail(T4D_ERR_ARG, "shs given but sh_coeffs == 0");
    T4DProblem p = *prob;
    if (!io->shs) p.sh_coeffs = 0;
    const Layout L = make_layout(p);
    if (io->state_bytes < L.total) return fail(T4D_ERR_STATE_SIZE, "state buffer smaller than t4d_state_bytes()");
    hipStream_t stream = (hipStream_t)hip_stream;
    const bool debug = (p.flags & T4D_FLAG_DEBUG_SYNC) != 0;
    const bool checked = debug || (p.flags & T4D_FLAG_CHECKED) != 0;
    char *st = (char *)io->state;

    KP kp;
    memset(&kp, 0, sizeof(kp));
    fill_common(kp, p, L, st);
    kp.views = io->views; kp.means3D = io->means3D; kp.opacities = io->opacities; kp.scales = io->scales;
    kp.rotations = io->rotations; kp.cov3D_precomp = io->cov3D_precomp; kp.colors_precomp = io->colors_precomp;
    kp.shs = io->shs;
    kp.out_color = io->out_color; kp.out_depth = io->out_depth; kp.out_alpha = io->out_alpha; kp.radii = io->out_radii;

    // un-synchronised one-view call with a pinned status block: the binning kernel writes it itself (publish_status)
    bool status_published = false;
    if (!checked && (p.flags & T4D_FLAG_ASYNC_STATUS) && status && p.n_views == 1 && kp.T <= kSmallTiles &&
        getenv("T4D_NO_SMALL_VIEW") == nullptr && getenv("T4D_STATUS_BY_COPY") == nullptr) {
        void *dptr = nullptr;
        if (hipHostGetDevicePointer(&dptr, (void *)status, 0) == hipSuccess && dptr != nullptr) {
            kp.host_status = (unsigned long long *)dptr;
            status_published = true;
        } else {
            (void)hipGetLastError();
        }
    }
    T4D_HIP(hipMemsetAsync(st, 0, L.zero_end, stream));
    // one view of at most 1,024 tiles (Topo4D's own call shape): scan and scatter are ONE launch (k_scan_scatter_small), and with
    // at most 128 workgroups of Gaussians (all resident at once) preprocess joins them behind a grid-wide barrier (k_front_small)
    const bool small_view = p.n_views == 1 && kp.T <= kSmallTiles && getenv("T4D_NO_SMALL_VIEW") == nullptr;
    const bool front = small_view && gaussian_grid(p.P, 1) + 1u <= 128u && getenv("T4D_NO_FRONT_FUSION") == nullptr;
    { ProfScope ps_(stream, K_PREPROCESS);
    if (front) hipLaunchKernelGGL(k_front_small, dim3(gaussian_grid(p.P, 1) + 1), dim3(kBlock), 0, stream, kp);
    else hipLaunchKernelGGL(k_preprocess, dim3(gaussian_grid(p.P, p.n_views)), dim3(kBlock), 0, stream, kp);
    }
    T4D_LAUNCH_CHECK("k_preprocess");
    if (front) {
    } else if (small_view) {
        ProfScope ps_(stream, K_SCATTER);
        hipLaunchKernelGGL(k_scan_scatter_small, dim3(gaussian_grid(p.P, 1) + 1), dim3(kBlock), 0, stream, kp);
    } else {
        ProfScope ps_(stream, K_SCAN_TILES);
        if (kp.n_chunks > 1) hipLaunchKernelGGL(k_tile_chunk_sums, dim3(kp.n_chunks, p.n_views), dim3(kScanChunk), 0, stream, kp);
        hipLaunchKernelGGL(k_scan_tiles, dim3(kp.n_chunks, p.n_views), dim3(kScanChunk), 0, stream, kp);
    }
    T4D_LAUNCH_CHECK("k_scan_tiles");
    if (checked) {
        DevStatus hs;
        T4D_HIP(hipMemcpyAsync(&hs, st + L.status, sizeof(hs), hipMemcpyDeviceToHost, stream));
        T4D_HIP(hipStreamSynchronize(stream));
        if (status) {
            status->max_pairs_per_view = hs.max_pairs;
            status->total_pairs = (int64_t)hs.total_pairs;
            status->overflow = (int32_t)hs.overflow;
            status->max_tile_pairs = (int32_t)min(hs.max_tile_pairs, 0x7fffffffu);
        }
        if (hs.overflow) return fail(T4D_ERR_PAIR_OVERFLOW, "pair_capacity too small for this scene");
    } else if ((p.flags & T4D_FLAG_ASYNC_STATUS) && status && !status_published) {
        // no synchronisation: the 16-byte raw status block lands in the caller's PINNED host memory once the scan kernel
        // has run; the caller looks at it after an event of its own (topo4d_amd's "auto" sync mode does, one call later)
        T4D_HIP(hipMemcpyAsync((void *)status, st + L.status, 16, hipMemcpyDeviceToHost, stream));     // the documented 16 bytes
    }
    if (!small_view) {
        ProfScope ps_(stream, K_SCATTER);
        hipLaunchKernelGGL(k_scatter, dim3(gaussian_grid(p.P, p.n_views) + (unsigned)(((size_t)kp.T * p.n_views + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, kp);
    }
    T4D_LAUNCH_CHECK("k_scatter");
    // Who sorts the bins.  A big launch: k_sort_tiles on 256 threads.  A small launch (at most kSegMaxTiles tiles) waits for its
    // longest bin: 1024 threads per bin - unless the forward runs its latency build AND the caller knows that every bin fits the
    // one-pass ranking sort (T4D_FLAG_SHORT_BINS): then the render workgroup of a tile sorts its own bin (no launch at all).
    const bool lat = latency_launch_fwd(kp.T * p.n_views, p.flags);
    kp.fused_sort = (lat && (p.flags & T4D_FLAG_SHORT_BINS) != 0 && getenv("T4D_NO_FUSED_SORT") == nullptr) ? 1u : 0u;
    if (!kp.fused_sort) {
        ProfScope ps_(stream, K_SORT_TILES);
        // (1024 threads per bin only pay when some bin is long: with the caller's word that every bin fits the ranking sort, a
        // small launch of several views keeps the 256-thread kernel - 4 views of Topo4D's size: 9.7 against 12.6 us)
        if (kp.slots_per_view != 0u && (p.flags & T4D_FLAG_SHORT_BINS) == 0 && getenv("T4D_SORT_256") == nullptr)
            hipLaunchKernelGGL(k_sort_tiles<kLongBlock>, dim3(min(kp.T * p.n_views, 4 * device_cus())), dim3(kLongBlock), 0, stream, kp);
        else
            hipLaunchKernelGGL(k_sort_tiles<kBlock>, dim3(tile_grid(kp.T * p.n_views, 5, 2)), dim3(kBlock), 0, stream, kp);
    }
    if (kp.long_bins_elsewhere)
        hipLaunchKernelGGL(k_sort_long, dim3(min(kp.T * p.n_views, device_cus())), dim3(kLongBlock), 0, stream, kp);
    T4D_LAUNCH_CHECK("k_sort_tiles");
    { ProfScope ps_(stream, K_RENDER_FWD);
    kp.tile_blocks = (uint32_t)(lat ? kp.T * p.n_views : tile_grid(kp.T * p.n_views, 6, 2));
    kp.fill_blocks = (uint32_t)(kp.gy * p.n_views);
    kp.fill_vec = (p.W % 4 == 0 && (((uintptr_t)io->out_color | (uintptr_t)io->out_depth | (uintptr_t)io->out_alpha) & 15u) == 0) ? 1u : 0u;
    if (getenv("T4D_FILL_SCALAR")) kp.fill_vec = 0u;       // tests: the 4-byte path on images that would take the 16-byte one
    const dim3 fgrid(kp.tile_blocks + kp.fill_blocks);
    const bool seg = kp.slots_per_view != 0u;        // small launch: snapshots for the segmented backward (kSeg)
    if (lat) {
        if (seg) hipLaunchKernelGGL((k_render_fwd<true, kBlock, true, true>), fgrid, dim3(kBlock), 0, stream, kp);
        else hipLaunchKernelGGL((k_render_fwd<true, kBlock, false, true>), fgrid, dim3(kBlock), 0, stream, kp);
    } else if (seg) {
        hipLaunchKernelGGL((k_render_fwd<false, kBlock, true, true>), fgrid, dim3(kBlock), 0, stream, kp);
    } else if (kp.long_bins_elsewhere && getenv("T4D_NO_PRUNE") == nullptr) {
        // a big launch that may hold long lists (the caller has not passed T4D_FLAG_NO_LONG_BINS): a dense pass
        hipLaunchKernelGGL((k_render_fwd<false, kFwdBatch, false, true>), fgrid, dim3(kBlock), 0, stream, kp);
    } else {
        hipLaunchKernelGGL((k_render_fwd<false, kFwdBatch, false, false>), fgrid, dim3(kBlock), 0, stream, kp);
    }
    }
    T4D_LAUNCH_CHECK("k_render_fwd");
    return T4D_OK;
}

T4D_EXPORT int t4d_rasterize_backward(const T4DProblem *prob, const T4DBackwardIO *io, void *hip_stream)
{
    int rc = check_problem(prob);
    if (rc != T4D_OK) return rc;
    if (!io || !io->views || !io->means3D || !io->opacities || !io->radii || !io->state || !io->dL_dcolor ||
        !io->dL_dmeans3D || !io->dL_dmeans2D || !io->dL_dopacities || !io->scratch)
        return fail(T4D_ERR_ARG, "null required pointer in T4DBackwardIO");
    if ((io->shs == nullptr) == (io->colors_precomp == nullptr))
        return fail(T4D_ERR_ARG, "provide exactly one of shs / colors_precomp");
    if (io->cov3D_precomp ? (io->scales || io->rotations) : (!io->scales || !io->rotations))
        return fail(T4D_ERR_ARG, "provide exactly one of (scales, rotations) / cov3D_precomp");
    if (io->shs ? !io->dL_dshs : !io->dL_dcolors) return fail(T4D_ERR_ARG, "missing colour gradient output");
    if (io->cov3D_precomp ? !io->dL_dcov3D : (!io->dL_dscales || !io->dL_drotations))
        return fail(T4D_ERR_ARG, "missing covariance gradient output");
    T4DProblem p = *prob;
    if (!io->shs) p.sh_coeffs = 0;
    const Layout L = make_layout(p);
    if (io->state_bytes < L.total) return fail(T4D_ERR_STATE_SIZE, "state buffer smaller than t4d_state_bytes()");
    if (io->scratch_bytes < t4d_backward_scratch_bytes(&p))
        return fail(T4D_ERR_STATE_SIZE, "scratch smaller than t4d_backward_scratch_bytes()");
    hipStream_t stream = (hipStream_t)hip_stream;
    const bool debug = (p.flags & T4D_FLAG_DEBUG_SYNC) != 0;
    char *st = (char *)io->state;

    KP kp;
    memset(&kp, 0, sizeof(kp));
    fill_common(kp, p, L, st);
    kp.views = io->views; kp.means3D = io->means3D; kp.opacities = io->opacities; kp.scales = io->scales;
    kp.rotations = io->rotations; kp.cov3D_precomp = io->cov3D_precomp; kp.colors_precomp = io->colors_precomp;
    kp.shs = io->shs;
    kp.radii = const_cast<int32_t *>(io->radii);
    kp.dL_dcolor = io->dL_dcolor; kp.dL_ddepth = io->dL_ddepth; kp.dL_dalpha = io->dL_dalpha;
    kp.grad_pair = (float *)io->scratch;
    if (io->cotangent_dot) {
        kp.tile_dot = (float *)((char *)io->scratch + grad_pair_bytes(p));
        kp.cotangent_dot = io->cotangent_dot;
    }
    kp.dL_dmeans3D = io->dL_dmeans3D; kp.dL_dmeans2D = io->dL_dmeans2D; kp.dL_dcolors = io->dL_dcolors;
    kp.dL_dshs = io->dL_dshs; kp.dL_dopacities = io->dL_dopacities; kp.dL_dscales = io->dL_dscales;
    kp.dL_drotations = io->dL_drotations; kp.dL_dcov3D = io->dL_dcov3D;

    { ProfScope ps_(stream, K_RENDER_BWD);
    const bool da = kp.dL_ddepth || kp.dL_dalpha;
    const bool lat = latency_launch(kp.T * p.n_views);
    // small launches: one workgroup per segment slot (kSeg; T4D_NO_SEGMENTS=1: whole tiles, for tests and experiments - the
    // forward's state serves both)
    const bool seg = kp.slots_per_view != 0u && getenv("T4D_NO_SEGMENTS") == nullptr;
    kp.tile_blocks = seg ? (uint32_t)p.n_views * kp.slots_per_view
                         : (uint32_t)(lat ? kp.T * p.n_views : tile_grid(kp.T * p.n_views, 4, 2));
    const uint32_t grid = kp.tile_blocks + (kp.tile_dot ? (uint32_t)p.n_views * ((kp.T + kEmptySpan - 1) / kEmptySpan) : 0u);
#define T4D_BWD_LAUNCH(DA_, LAT_, SEG_) hipLaunchKernelGGL((k_render_bwd<DA_, LAT_, SEG_>), dim3(grid), dim3(kBlock), 0, stream, kp)
    if (seg) {
        // Segments always run the throughput build: the latency build's one slab per DPP row is 82 KiB of LDS, ONE workgroup per
        // CU, and a one-view launch has more segments than CUs (432 at Topo4D's size: two rounds, 47 us against 29 us measured)
        if (da) T4D_BWD_LAUNCH(true, false, true); else T4D_BWD_LAUNCH(false, false, true);
    } else {
        if (lat) { if (da) T4D_BWD_LAUNCH(true, true, false); else T4D_BWD_LAUNCH(false, true, false); }
        else { if (da) T4D_BWD_LAUNCH(true, false, false); else T4D_BWD_LAUNCH(false, false, false); }
    }
#undef T4D_BWD_LAUNCH
    }
    T4D_LAUNCH_CHECK("k_render_bwd");
    { ProfScope ps_(stream, K_PREPROCESS_BWD);
    const dim3 pgrid(gaussian_grid(p.P, p.n_views) + (kp.tile_dot ? p.n_views : 0));
    if (kp.shs)          // SH colours: the per-Gaussian kernel leaves dL/dcolour in the scratch, k_sh_bwd takes it from there
        kp.dL_dcolors = (float *)((char *)io->scratch + grad_pair_bytes(p) + tile_dot_bytes(p, (size_t)kp.T));
    hipLaunchKernelGGL(k_preprocess_bwd, pgrid, dim3(kBlock), 0, stream, kp);
    if (kp.shs)
    {
        const bool plain = getenv("T4D_SH_BWD_PLAIN") != nullptr;                // tests / experiments: the general kernel also for degree 3
        if (kp.M == 16 && kp.deg == 3 && !plain)
            hipLaunchKernelGGL(k_sh_bwd16, dim3(gaussian_grid(p.P, (p.n_views + T4D_SHB_VIEWS - 1) / T4D_SHB_VIEWS)), dim3(kBlock), 0, stream, kp);
        else
            hipLaunchKernelGGL(k_sh_bwd, dim3(gaussian_grid(p.P, p.n_views)), dim3(kBlock), 0, stream, kp);
    }
    }
    T4D_LAUNCH_CHECK("k_preprocess_bwd");
    return T4D_OK;
}

#ifdef T4D_COUNT
T4D_EXPORT int t4d_debug_read_counters(unsigned long long *out, int reset)
{
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_count), sizeof(unsigned long long) * 16);
    if (e == hipSuccess && reset) {
        const unsigned long long z[16] = { 0 };
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_count), z, sizeof(z));
    }
    return (int)e;
}
#endif


T4D_EXPORT int t4d_profile_begin(void)
{
    for (auto &r : g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_prof.clear();
    g_prof_on = true;
    return T4D_OK;
}

T4D_EXPORT int t4d_profile_end(T4DKernelTime *out, int max_entries, int *n_entries)
{
    g_prof_on = false;
    if (!out || !n_entries || max_entries < K_COUNT) return fail(T4D_ERR_ARG, "need room for every kernel");
    for (int k = 0; k < K_COUNT; k++) {
        out[k].name = kKernelNames[k];
        out[k].total_ms = 0.0;
        out[k].launches = 0;
    }
    for (auto &r : g_prof) {
        T4D_HIP(hipEventSynchronize(r.b));
        float ms = 0.f;
        T4D_HIP(hipEventElapsedTime(&ms, r.a, r.b));
        out[r.id].total_ms += ms;
        out[r.id].launches += 1;
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    g_prof.clear();
    *n_entries = K_COUNT;
    return T4D_OK;
}

T4D_EXPORT int t4d_fetch_status(const T4DProblem *prob, const void *state, T4DStatus *out, void *hip_stream)
{
    int rc = check_problem(prob);
    if (rc != T4D_OK) return rc;
    if (!state || !out) return fail(T4D_ERR_ARG, "null pointer");
    hipStream_t stream = (hipStream_t)hip_stream;
    DevStatus hs;
    T4D_HIP(hipMemcpyAsync(&hs, state, sizeof(hs), hipMemcpyDeviceToHost, stream));
    T4D_HIP(hipStreamSynchronize(stream));
    out->max_pairs_per_view = hs.max_pairs;
    out->total_pairs = (int64_t)hs.total_pairs;
    out->overflow = (int32_t)hs.overflow;
    out->max_tile_pairs = (int32_t)min(hs.max_tile_pairs, 0x7fffffffu);
    return T4D_OK;
}

T4D_EXPORT size_t t4d_view_dot_scratch_bytes(int32_t n_views) { return n_views > 0 ? (size_t)n_views * kDotBlocks * sizeof(float) : 0; }

T4D_EXPORT int t4d_view_dot(int32_t n_views, int64_t n_per_view, const float *a, const float *b, float *out, void *scratch,
                            void *hip_stream)
{
    if (n_views < 1 || n_per_view < 1 || !a || !b || !out || !scratch) return fail(T4D_ERR_ARG, "bad arguments");
    hipStream_t stream = (hipStream_t)hip_stream;
    const bool debug = false;
    hipLaunchKernelGGL(k_view_dot_partial, dim3(kDotBlocks, n_views), dim3(kBlock), 0, stream, a, b, (size_t)n_per_view,
                       (float *)scratch);
    T4D_LAUNCH_CHECK("k_view_dot_partial");
    hipLaunchKernelGGL(k_view_dot_final, dim3(n_views), dim3(64), 0, stream, (const float *)scratch, out);
    T4D_LAUNCH_CHECK("k_view_dot_final");
    return T4D_OK;
}

T4D_EXPORT int t4d_mark_visible(int32_t P, const float *means3D, const float *view, uint8_t *present, void *hip_stream)
{
    if (P < 0 || (P > 0 && (!means3D || !view || !present))) return fail(T4D_ERR_ARG, "bad arguments");
    if (P == 0) return T4D_OK;
    hipStream_t stream = (hipStream_t)hip_stream;
    const bool debug = false;
    hipLaunchKernelGGL(k_mark_visible, dim3((P + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, P, means3D, view, present);
    T4D_LAUNCH_CHECK("k_mark_visible");
    return T4D_OK;
}
