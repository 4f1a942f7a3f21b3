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
// One translation unit in six files: this one (constants, state layout, kernel parameter block, device helpers, host side and C ABI)
// and, included below, t4d_raster_binning.h, t4d_raster_sort.h, t4d_raster_render_fwd.h, t4d_raster_render_bwd.h and
// t4d_raster_gaussian_bwd.h.
// Kernels (DESIGN.md has the bytes/roofline of each):
//   k_preprocess      A.1  per (view,Gaussian): cull, project, cov3D, EWA cov2D, conic, radius, tile rect, SH colour;
//                          + per-tile counts and pair ranks + pair-slot allocation (one returning atomic per workgroup)
//   k_tile_chunk_sums, k_scan_tiles  A.2  per (view, 1024-tile chunk): exclusive scan of tile counts -> tile offsets, overflow
//                          status, longest list, length buckets (the first kernel only when a view has more than one chunk)
//   k_scatter         A.2  per (view,Gaussian): key -> tile_off + rank (no atomics); tail blocks build the work items
//   k_sort_tiles      A.2  per work item: sort the bin by (depth bits, index): runs of 64 sorted in registers, merged by
//                          ranking (<= 512 keys: one pass; <= 2048: log levels in LDS)
//   k_sort_long_chunks, k_merge_long  A.2  bins beyond 2048 keys (dense passes): every 2048-key chunk sorted through LDS by whichever
//                          workgroup it falls to, then ONE ranking pass (a key's slot = its place in its chunk + the keys below it in
//                          every other chunk) scatters the keys to their final places - any bin length, chunks in parallel
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
#include "t4d_activations.h"

#define T4D_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

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
constexpr int kLongBlock = 1024;     // k_sort_long_chunks, k_merge_long (bins beyond kSortLdsCap keys) and small launches' k_sort_tiles: threads per workgroup
constexpr int kScanChunk = 1024;     // tiles scanned per workgroup of k_scan_tiles
constexpr int kRankSortMax = 512;    // bins up to this length: register-sorted runs of 64 + one ranking pass (one barrier)
constexpr int kHist = 1024;          // per-workgroup LDS tile histogram (bounding box of the tiles a workgroup touches)
constexpr int kBuckets = 24;         // tile-length classes (floor(log2 n), descending; last = empty) for launch ordering
constexpr int kGP = T4D_GRAD_PAIR_FLOATS;
// pair-slot cursors per view, at most (same-address returning atomics are serial: see k_preprocess).  A view gets one cursor per 16
// workgroups of Gaussians up to this many: 8 at config 2 (118 workgroups per view), 32 at config 4, 64 for ONE view of 10^6 Gaussians -
// 3,907 workgroups on 8 cursors were 488 serial atomics each: k_preprocess 60.4 -> 37.8 us there (round 5; 256 cursors: 33.5, but the
// scan's walk over them costs what that saves)
constexpr int kCursorSegs = 64;
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
constexpr int kSeg = T4D_SEG;        // list positions per backward segment of a 2-8 view launch; a ONE-view launch takes kSegOne (seg_positions)
constexpr int kSegOne = 64;          // round 4 measured it with a compile-time switch: one view of Topo4D's size 85.4 -> 81.3 us (half the walk per
                                     // work item, twice the list rounds in the forward), three views of the config-2 scene 176 -> 183: hence per launch
constexpr int kSegMaxTiles = 8192;   // launches of at most this many tiles (V * T) run the segmented backward
// A ONE-view launch of more tiles than that (the texture pass: one 4096 x 3008 view = 48,128 tiles of 10^6 Gaussians) fills the chip
// with whole tiles - until the short tiles run out and a few workgroups are still walking lists of thousands of pairs: the SAME
// number of wave-steps (counting build) took the backward 343 us on a scene whose longest list held 1,459 pairs and 400 / 906 us
// (two cameras) on one with lists of 9,000 - 12,000.  There the tiles of at least kSegLongMin pairs - and only those - are cut into
// segments (kSeg positions each, their own launch behind the whole-tile one, which skips them): same box, whole tiles / threshold
// 1,024 / 2,048 / 4,096: 400 / 406 / 403 / 382 us and 906 / 414 / 398 / 529 us.  (Segments for EVERY tile of such a launch, the
// small launches' way: 488 / 506 us and 50 us more in the forward, which then keeps 40,000 snapshots.)
#ifndef T4D_SEG_LONG_MIN
#define T4D_SEG_LONG_MIN 2048
#endif
constexpr int kSegLongMin = T4D_SEG_LONG_MIN;
constexpr int kSnapFloats = 6;       // T, C0, C1, C2, D per pixel and boundary (+ the last contributor so far: the depth-parallel forward of long tiles)
static_assert((kSegLongMin & (kSegLongMin - 1)) == 0, "k_fwd_long_prefix stops at the first work item of a shorter length CLASS (powers of two)");

thread_local char g_err[512] = "";

// optional per-kernel timing with HIP events (t4d_profile_begin/end); used by bench.py for the roofline object
// (the last three only run for launches that may hold long tile lists - a dense pass: the chunk sort + merge of bins beyond the LDS
// sort buffer, the three depth-parallel forward launches and the segmented backward of a big one-view launch's long tiles)
enum KernelId { K_PREPROCESS = 0, K_SCAN_TILES, K_SCATTER, K_SORT_TILES, K_RENDER_FWD, K_RENDER_BWD, K_PREPROCESS_BWD,
                K_SORT_LONG, K_FWD_LONG, K_RENDER_BWD_LONG, K_COUNT };
const char *const kKernelNames[K_COUNT] = { "k_preprocess", "k_scan_tiles", "k_scatter", "k_sort_tiles", "k_render_fwd",
                                            "k_render_bwd", "k_preprocess_bwd", "k_sort_long", "k_fwd_long", "k_render_bwd_long" };
struct ProfRec { int id; hipEvent_t a, b; };
bool g_prof_on = false;
std::vector<ProfRec> g_prof;

// ---------------------------------------------------------------------------------------------------------
// state / scratch layout
// ---------------------------------------------------------------------------------------------------------
struct Layout {
    size_t status, view_total, view_cursor, tile_count, bucket_fill, slot_tab, zero_end;
    size_t snap, live;
    size_t tile_off, chunk_sum, order, items, xy, depth, conic_opacity, rgb, clamped, pair_off, pair_rank, keys, sort_tmp, final_T, n_contrib, total;
};

struct DevStatus {            // first bytes of the state buffer
    uint32_t overflow;
    uint32_t max_pairs;
    unsigned long long total_pairs;       // ... the 16 bytes T4D_FLAG_ASYNC_STATUS copies out end here
    uint32_t max_tile_pairs;              // longest tile list of the call (reported as T4DStatus.max_tile_pairs)
    uint32_t grid_sync;                   // arrival counter of k_front_small's one grid-wide barrier
    uint32_t live_segments;               // seg_mode 2: entries of the live-segment list (the long tiles' segments, Layout::live)
};

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// segmented backward (see kSeg): decided by the problem's dimensions alone, so that t4d_state_bytes, the forward and the
// backward of a call agree without talking to each other
// 0: whole tiles only; 1: every tile is segmented (small launch); 2: the long tiles of a big one-view launch are
inline int seg_mode(const T4DProblem &p)
{
    const long long T = (long long)((p.W + T4D_TILE_X - 1) / T4D_TILE_X) * (long long)((p.H + T4D_TILE_Y - 1) / T4D_TILE_Y);
    if ((long long)p.n_views * T <= kSegMaxTiles) return 1;
    return p.n_views == 1 ? 2 : 0;
}
inline bool seg_capable(const T4DProblem &p) { return seg_mode(p) != 0; }
// Segment slots of a view.  Tile t (arena offset off, n pairs) owns the slots floor(off / kSeg) + t ... + ceil(n / kSeg) - 1:
// disjoint from tile to tile ((off + n) / kSeg - off / kSeg >= floor(n / kSeg)) without a prefix sum over the tiles.
// seg_mode 2 (only the tiles of at least kSegLongMin pairs own slots): floor(off / kSeg) + floor(off / kSegLongMin) instead - the
// second term grows by at least one from a long tile to the next ((off + n) / kSegLongMin - off / kSegLongMin >= 1 for n >=
// kSegLongMin), which is all the "+ t" was for: cap / kSeg + cap / kSegLongMin slots instead of cap / kSeg + T (one
// 4096 x 3008 view carried 48,128 x 6 KB = 290 MB of snapshot slots for tiles that can never own one).
inline int seg_positions(const T4DProblem &p) { return (p.n_views == 1 && seg_mode(p) == 1) ? kSegOne : kSeg; }
inline size_t seg_slots_per_view(const T4DProblem &p, size_t T)
{
    if (seg_mode(p) == 2) return (size_t)p.pair_capacity / (size_t)kSeg + (size_t)p.pair_capacity / (size_t)kSegLongMin + 2;
    return (size_t)p.pair_capacity / (size_t)seg_positions(p) + T + 1;
}

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
    // seg_mode 2: the slots that hold a segment, compact (the table itself is mostly empty: a few thousand segments in 60,000 slots)
    L.live = o;          o = align_up(o + (seg_mode(p) == 2 ? slots * 4 : 0));
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
    uint32_t seg_shift;              // log2 of the backward's segment length of this launch (seg_positions: 6 or 7)
    const float *views, *means3D, *opacities, *scales, *rotations, *cov3D_precomp, *colors_precomp, *shs;
    // state
    DevStatus *status;
    uint32_t *view_total, *view_cursor, *tile_count, *bucket_fill, *tile_off, *chunk_sum, *order, *pair_off, *pair_rank;
    int n_chunks;                    // scan chunks per view = ceil(T / kScanChunk)
    int long_bins_elsewhere;         // 1: k_sort_long runs behind k_sort_tiles and takes the bins longer than kSortLdsCap
    int raw_params;                  // T4D_FLAG_RAW_PARAMS: rotations / opacities / scales are the optimiser's raw parameters
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
    // (behind everything else: the kernels of every other launch shape read their arguments from the offsets they always had)
    uint32_t seg_min_pairs;          // tiles of fewer pairs are not segmented (0: every tile is - small launches; kSegLongMin: seg_mode 2)
    uint32_t *live;                  // seg_mode 2: slot-table indices of the live segments, status->live_segments of them (order: as the item builders got to them)
    uint32_t slots_by_offset;        // seg_mode 2: a long tile's first slot is off / kSeg + off / kSegLongMin (seg_slot0), not off / kSeg + tile
    uint32_t views_per_set;          // T4DProblem.views_per_param_set: view v reads the per-Gaussian inputs of parameter set v / views_per_set (0: one set for all views)
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

// First row of view v's parameter set in the per-Gaussian INPUT arrays (means3D, opacities, scales, rotations, cov3D_precomp,
// colors_precomp, shs): a launch may carry the views of several frames - T4DProblem.views_per_param_set consecutive views per
// frame, the frames' parameters one [P, .] block after the other.  v is workgroup-uniform everywhere: one scalar division.
__device__ __forceinline__ size_t param_row0(const KP &kp, const int v)
{
    return kp.views_per_set ? (size_t)((uint32_t)v / kp.views_per_set) * (size_t)kp.P : (size_t)0;
}

// first segment slot (within its view) of the tile at arena offset `off`: see seg_slots_per_view
__device__ __forceinline__ uint32_t seg_slot0(const KP &kp, const uint32_t off, const uint32_t tile)
{
    static_assert((kSegLongMin & (kSegLongMin - 1)) == 0, "a shift");
    return (off >> kp.seg_shift) + (kp.slots_by_offset ? off / (uint32_t)kSegLongMin : tile);
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
// view-fastest order of round 2) or once per view (block-fastest).  k_preprocess_bwd and the SH backward kernels take this order;
// k_preprocess does not: it is a third SLOWER with it (211 -> 288 us at config 4) although its fetch traffic falls.
__device__ __forceinline__ bool block_and_view(const uint32_t b, const uint32_t V, const uint32_t nblocks, uint32_t &gb, uint32_t &v)
{
    const uint32_t per = 8u * V, grp = b / per, rem = b - grp * per;
    v = rem >> 3;
    gb = grp * 8u + (rem & 7u);
    return gb < nblocks;
}
__host__ __device__ inline unsigned gaussian_grid(const int P, const int V) { return (unsigned)((((P + kBlock - 1) / kBlock + 7) / 8) * 8 * V); }

// ---------------------------------------------------------------------------------------------------------
// the kernels, one file per stage of the path (all part of THIS translation unit and of its anonymous namespace)
// ---------------------------------------------------------------------------------------------------------
#include "t4d_raster_binning.h"
#include "t4d_raster_sort.h"
#include "t4d_raster_render_fwd.h"
#include "t4d_raster_render_fwd_long.h"
#include "t4d_raster_render_bwd.h"
#include "t4d_raster_gaussian_bwd.h"

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
    if (p->views_per_param_set != 0u && (p->views_per_param_set > (uint32_t)p->n_views || (uint32_t)p->n_views % p->views_per_param_set != 0u))
        return fail(T4D_ERR_ARG, "n_views must be a multiple of views_per_param_set");
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
    kp.raw_params = (p.flags & T4D_FLAG_RAW_PARAMS) ? 1 : 0;
    kp.views_per_set = (p.views_per_param_set != 0u && p.views_per_param_set < (uint32_t)p.n_views) ? p.views_per_param_set : 0u;
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
    kp.live = reinterpret_cast<uint32_t *>(st + L.live);
    kp.slots_per_view = seg_capable(p) ? (uint32_t)seg_slots_per_view(p, (size_t)kp.T) : 0u;
    kp.seg_shift = seg_positions(p) == 64 ? 6u : 7u;
    kp.slots_by_offset = seg_mode(p) == 2 ? 1u : 0u;
    // seg_mode 2: the caller's word that no list is long (T4D_FLAG_NO_LONG_BINS) keeps every tile whole
    kp.seg_min_pairs = seg_mode(p) == 2 ? ((p.flags & T4D_FLAG_NO_LONG_BINS) ? 0xffffffffu : (uint32_t)kSegLongMin) : 0u;
    static_assert(kSegOne == 64 && kSeg == 128, "seg_shift assumes segment lengths of 64 and 128");
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

// how many workgroups of k_front_small the current device holds at once (occupancy x compute units; cached per device)
static unsigned front_small_resident_blocks()
{
    static int cached_dev = -1;
    static unsigned cached = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    if (dev != cached_dev) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_front_small, kBlock, 0) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || per_cu < 1 || cus < 1) {
            (void)hipGetLastError();
            per_cu = 0; cus = 0;                         // unknown: never take the barrier kernel
        }
        cached = (unsigned)per_cu * (unsigned)cus;
        cached_dev = dev;
    }
    return cached;
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
    // (the grid barrier is only safe when every workgroup of the launch is resident at once: checked against what the DEVICE can
    // hold - a CU mask or a compute partition of 32 CUs shows up in device_cus() - not assumed from the workgroup count)
    const bool front = small_view && gaussian_grid(p.P, 1) + 1u <= 128u && gaussian_grid(p.P, 1) + 1u <= front_small_resident_blocks() &&
                       getenv("T4D_NO_FRONT_FUSION") == nullptr;
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
    // a big one-view launch that may hold long lists: its long tiles go through the depth-parallel kernels, and the throughput forward
    // that renders the others sorts its own tiles' bins first (k_render_fwd, FUSE): no k_sort_tiles launch
    const bool long_fwd = kp.slots_per_view != 0u && kp.seg_min_pairs != 0xffffffffu && seg_mode(p) == 2 && !lat && getenv("T4D_NO_LONG_FWD") == nullptr;
    if (long_fwd && getenv("T4D_NO_FUSED_SORT") == nullptr) kp.fused_sort = 1u;
    if (!kp.fused_sort) {
        ProfScope ps_(stream, K_SORT_TILES);
        // (1024 threads per bin only pay when some bin is long: with the caller's word that every bin fits the ranking sort, a
        // small launch of several views keeps the 256-thread kernel - 4 views of Topo4D's size: 9.7 against 12.6 us)
        if (seg_mode(p) == 1 && (p.flags & T4D_FLAG_SHORT_BINS) == 0 && getenv("T4D_SORT_256") == nullptr)
            hipLaunchKernelGGL(k_sort_tiles<kLongBlock>, dim3(min(kp.T * p.n_views, 4 * device_cus())), dim3(kLongBlock), 0, stream, kp);
        else
            hipLaunchKernelGGL(k_sort_tiles<kBlock>, dim3(tile_grid(kp.T * p.n_views, 5, 2)), dim3(kBlock), 0, stream, kp);
    }
    T4D_LAUNCH_CHECK("k_sort_tiles");
    // bins beyond the LDS sort buffer (the caller has not said that there are none): their chunks, then the merges
    if (kp.long_bins_elsewhere) {
        ProfScope ps_(stream, K_SORT_LONG);
        hipLaunchKernelGGL(k_sort_long_chunks, dim3(min(kp.T * p.n_views, 2 * device_cus())), dim3(kLongBlock), 0, stream, kp);
        hipLaunchKernelGGL(k_merge_long, dim3(min(kp.T * p.n_views, 2 * device_cus())), dim3(kLongBlock), 0, stream, kp);
    }
    T4D_LAUNCH_CHECK("k_sort_long");
    {
    kp.tile_blocks = (uint32_t)(lat ? kp.T * p.n_views : tile_grid(kp.T * p.n_views, 6, 2));
    kp.fill_blocks = (uint32_t)(kp.gy * p.n_views);
    kp.fill_vec = (p.W % 4 == 0 && (((uintptr_t)io->out_color | (uintptr_t)io->out_depth | (uintptr_t)io->out_alpha) & 15u) == 0) ? 1u : 0u;
    if (getenv("T4D_FILL_SCALAR")) kp.fill_vec = 0u;       // tests: the 4-byte path on images that would take the 16-byte one
    const dim3 fgrid(kp.tile_blocks + kp.fill_blocks);
    // small launch, or a big one-view launch that may hold long lists: snapshots for the segmented backward (kSeg)
    const bool seg = kp.slots_per_view != 0u && kp.seg_min_pairs != 0xffffffffu;
    const bool seg_one = seg && seg_positions(p) == kSegOne;
    // a big one-view launch that may hold long lists: those tiles go through the depth-parallel kernels (three launches over the
    // slot table, t4d_raster_render_fwd_long.h); the others through the throughput build, which leaves the long ones out
    if (long_fwd) {
        { ProfScope ps_(stream, K_RENDER_FWD);
        hipLaunchKernelGGL((k_render_fwd<false, kFwdBatch, 0, true, true>), fgrid, dim3(kBlock), 0, stream, kp);
        }
        T4D_LAUNCH_CHECK("k_render_fwd");
        ProfScope psl_(stream, K_FWD_LONG);
        KP kl = kp;
        kl.tile_blocks = min(kp.slots_per_view, (uint32_t)(8 * device_cus()));
        hipLaunchKernelGGL(k_fwd_long_seg<false>, dim3(kl.tile_blocks), dim3(kBlock), 0, stream, kl);
        T4D_LAUNCH_CHECK("k_fwd_long_seg");
        hipLaunchKernelGGL(k_fwd_long_prefix, dim3(min((uint32_t)kp.T, (uint32_t)(4 * device_cus()))), dim3(kBlock), 0, stream, kl);
        T4D_LAUNCH_CHECK("k_fwd_long_prefix");
        hipLaunchKernelGGL(k_fwd_long_seg<true>, dim3(kl.tile_blocks), dim3(kBlock), 0, stream, kl);
    } else {
    ProfScope ps_(stream, K_RENDER_FWD);
    if (lat) {
        if (seg_one) hipLaunchKernelGGL((k_render_fwd<true, kBlock, kSegOne, true>), fgrid, dim3(kBlock), 0, stream, kp);
        else if (seg) hipLaunchKernelGGL((k_render_fwd<true, kBlock, kSeg, true>), fgrid, dim3(kBlock), 0, stream, kp);
        else hipLaunchKernelGGL((k_render_fwd<true, kBlock, 0, true>), fgrid, dim3(kBlock), 0, stream, kp);
    } else if (seg_one) {
        hipLaunchKernelGGL((k_render_fwd<false, kBlock, kSegOne, true>), fgrid, dim3(kBlock), 0, stream, kp);
    } else if (seg) {
        hipLaunchKernelGGL((k_render_fwd<false, kBlock, kSeg, true>), fgrid, dim3(kBlock), 0, stream, kp);
    } else if (kp.long_bins_elsewhere && getenv("T4D_NO_PRUNE") == nullptr) {
        // a big launch that may hold long lists (the caller has not passed T4D_FLAG_NO_LONG_BINS): a dense pass
        hipLaunchKernelGGL((k_render_fwd<false, kFwdBatch, 0, true>), fgrid, dim3(kBlock), 0, stream, kp);
    } else {
        hipLaunchKernelGGL((k_render_fwd<false, kFwdBatch, 0, false>), fgrid, dim3(kBlock), 0, stream, kp);
    }
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

    {
    const bool da = kp.dL_ddepth || kp.dL_dalpha;
    const bool lat = latency_launch(kp.T * p.n_views);
    // small launches: one workgroup per segment slot (kSeg; T4D_NO_SEGMENTS=1: whole tiles, for tests and experiments - the
    // forward's state serves both)
    // (seg_mode 2 - the long tiles of a big one-view launch - only when the caller has not said that there are none: the forward
    // of a call that says so keeps no snapshots, and a forward that kept them is simply not used)
    const bool seg_long = seg_mode(p) == 2 && kp.seg_min_pairs != 0xffffffffu && getenv("T4D_NO_SEGMENTS") == nullptr;
    const bool seg = seg_mode(p) == 1 && getenv("T4D_NO_SEGMENTS") == nullptr;
    kp.tile_blocks = seg ? (uint32_t)p.n_views * kp.slots_per_view
                         : (uint32_t)(lat ? kp.T * p.n_views : tile_grid(kp.T * p.n_views, 4, 2));
    uint32_t grid = kp.tile_blocks + (kp.tile_dot ? (uint32_t)p.n_views * ((kp.T + kEmptySpan - 1) / kEmptySpan) : 0u);
#define T4D_BWD_LAUNCH(DA_, LAT_, SEG_) hipLaunchKernelGGL((k_render_bwd<DA_, LAT_, SEG_>), dim3(grid), dim3(kBlock), 0, stream, kp)
    if (seg_long) {
        // two launches: the whole-tile throughput build leaves out the tiles that own segments (the forward wrote their slot-table
        // entries), then the segmented build walks the slot table with a fixed number of workgroups (most of its cap / kSeg + T
        // slots are empty: one workgroup per slot would be 79,000 launches for a few thousand segments)
        { ProfScope ps_(stream, K_RENDER_BWD);
        if (da) hipLaunchKernelGGL((k_render_bwd<true, false, 0, true>), dim3(grid), dim3(kBlock), 0, stream, kp);
        else hipLaunchKernelGGL((k_render_bwd<false, false, 0, true>), dim3(grid), dim3(kBlock), 0, stream, kp);
        }
        T4D_LAUNCH_CHECK("k_render_bwd");
        ProfScope psl_(stream, K_RENDER_BWD_LONG);
        kp.tile_blocks = min(kp.slots_per_view, (uint32_t)(8 * device_cus()));
        grid = kp.tile_blocks;
        if (da) hipLaunchKernelGGL((k_render_bwd<true, false, kSeg, true>), dim3(grid), dim3(kBlock), 0, stream, kp);
        else hipLaunchKernelGGL((k_render_bwd<false, false, kSeg, true>), dim3(grid), dim3(kBlock), 0, stream, kp);
    } else {
    ProfScope ps_(stream, K_RENDER_BWD);
    if (seg && seg_positions(p) == kSegOne) {
        if (da) T4D_BWD_LAUNCH(true, false, kSegOne); else T4D_BWD_LAUNCH(false, false, kSegOne);
    } else if (seg) {
        // Segments always run the throughput build: the latency build's one slab per DPP row is 82 KiB of LDS, ONE workgroup per
        // CU, and a one-view launch has more segments than CUs (432 at Topo4D's size: two rounds, 47 us against 29 us measured)
        if (da) T4D_BWD_LAUNCH(true, false, kSeg); else T4D_BWD_LAUNCH(false, false, kSeg);
    } else {
        if (lat) { if (da) T4D_BWD_LAUNCH(true, true, 0); else T4D_BWD_LAUNCH(false, true, 0); }
        else { if (da) T4D_BWD_LAUNCH(true, false, 0); else T4D_BWD_LAUNCH(false, false, 0); }
    }
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
        // (k_sh_bwd16 serves T4D_SHB_VIEWS views from one fetch of the coefficient rows: they must belong to one parameter set)
        if (kp.M == 16 && kp.deg == 3 && !plain && (kp.views_per_set == 0u || kp.views_per_set % T4D_SHB_VIEWS == 0u))
            hipLaunchKernelGGL(k_sh_bwd16, dim3(gaussian_grid(p.P, (p.n_views + T4D_SHB_VIEWS - 1) / T4D_SHB_VIEWS)), dim3(kBlock), 0, stream, kp);
        else
            hipLaunchKernelGGL(k_sh_bwd, dim3(gaussian_grid(p.P, p.n_views)), dim3(kBlock), 0, stream, kp);
    }
    }
    T4D_LAUNCH_CHECK("k_preprocess_bwd");
    return T4D_OK;
}



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
