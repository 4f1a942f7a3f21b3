// t4d_texture.hip — UV-space texture bake on MI355X (BASELINE config 5).
//
// Replaces, for Topo4D's per-frame texture export (reference helpers.py:953-960 `write_texture` ->
// face3d/mesh/render.py:52-86 `render_colors`), the single-threaded CPU rasterizer
// `_render_colors_core` (face3d/mesh/cython/mesh_core.cpp:169-234).  Results are BIT-IDENTICAL to that code
// (tests compare against the reference's own source compiled into oracle/_ref).
//
// The reference walks the triangles serially and z-tests with a strict `>`, so a texel ends up with the triangle
// of maximum interpolated depth and, among equals, the LOWEST index (for Topo4D every depth is 0: first triangle
// wins).  That final state is order-independent, which is what makes a parallel formulation exact:
//   k_tex_bin<count> / k_tex_scan / k_tex_bin<fill>   bin triangles by the 32x32-texel tiles their clipped pixel bbox touches
//                                           (LDS histogram per workgroup, one global atomic per touched tile; the scan runs in
//                                           1024-bin chunks on as many workgroups)
//   k_tex_render                            one workgroup per tile: like the reference, a triangle visits only the texels of
//                                           its bounding box (a wave per triangle, lanes over the box), and the texels keep the
//                                           lexicographic max of (depth, -index) through an LDS 64-bit integer maximum
// including the reference's quirk that texels in the 2-pixel border ring of the image are drawn from any triangle
// whose bbox contains them, with extrapolated barycentrics (mesh_core.cpp:211).
// All arithmetic is written in the reference's operation order with FP contraction off.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/topo4d_raster.h"

#define T4D_EXPORT extern "C" __attribute__((visibility("default")))
int t4d_internal_fail(int code, const char *fmt, const char *a);

namespace {

constexpr int kTile = 32;        // bin = workgroup tile: 32x32 texels, four per thread (the dependent-load chain of a workgroup -
                                 // bin header -> records -> colours - is then paid once per 1024 texels: at 16x16 the kernel was bound by it)
constexpr int kBlock = 256;
constexpr int kStage = 120;      // triangles set up in LDS per round: 120 x (64 + 36) B + 8 KiB of keys = 20.2 KB per workgroup, eight per CU

struct Tri {                                // 80 bytes = five 16-byte words: the unit of the per-tile lists
    float p0x, p0y, v0x, v0y, v1x, v1y;     // p0, v0 = p2 - p0, v1 = p1 - p0
    float dot00, dot01, dot11, inverDeno;
    float d0, d1, d2;
    int x_min, x_max, y_min, y_max;
    int idx;
    int i0, i1;                             // (pads the record to five 16-byte words)
};
static_assert(sizeof(Tri) == 80, "Tri must stay five float4 words");
constexpr int kScanChunk = 1024;
#ifndef T4D_TEX_GROUP
#define T4D_TEX_GROUP 16               // lanes per triangle record in the texel loop (same box: 64 lanes 0.679 ms, 32: 0.638, 16: 0.622, 8: 0.640)
#endif

struct TexP {
    const float *vertices;
    const int32_t *triangles;
    const float *colors;
    int nver, ntri, h, w, c, row_begin, row_end;
    int bx, by, by0;              // bins in x, bins in y inside the row band, first bin row of the band
    int n_chunks;
    uint32_t cap;
    uint32_t *bin_count, *bin_cursor, *bin_off, *chunk_sum;
    uint4 *list;                  // per (tile, triangle) pair: the triangle's index and its three vertex indices (one fetch level less per tile)
    unsigned long long *total;
    float *image, *depth;
    const float *bg;              // FRESH launches: background image [h,w,c] or nullptr = zeros
};

// pixel bbox exactly as mesh_core.cpp:190-199, additionally clipped to the row band
__device__ __forceinline__ bool tri_bbox(const TexP &P, const int i, int &x_min, int &x_max, int &y_min, int &y_max, int &i0, int &i1, int &i2)
{
    i0 = P.triangles[3 * (size_t)i]; i1 = P.triangles[3 * (size_t)i + 1]; i2 = P.triangles[3 * (size_t)i + 2];
    const float x0 = P.vertices[3 * (size_t)i0], y0 = P.vertices[3 * (size_t)i0 + 1];
    const float x1 = P.vertices[3 * (size_t)i1], y1 = P.vertices[3 * (size_t)i1 + 1];
    const float x2 = P.vertices[3 * (size_t)i2], y2 = P.vertices[3 * (size_t)i2 + 1];
    x_min = max((int)ceilf(fminf(x0, fminf(x1, x2))), 0);
    x_max = min((int)floorf(fmaxf(x0, fmaxf(x1, x2))), P.w - 1);
    y_min = max((int)ceilf(fminf(y0, fminf(y1, y2))), 0);
    y_max = min((int)floorf(fmaxf(y0, fmaxf(y1, y2))), P.h - 1);
    if (x_max < x_min || y_max < y_min) return false;
    return !(y_max < P.row_begin || y_min >= P.row_end);
}

// Counting and filling go through a per-workgroup LDS histogram over the bounding box (in tiles) of everything the workgroup's
// 256 triangles touch: triangles that are neighbours in the index buffer are neighbours in UV space, so a workgroup hits a
// handful of tiles and sends ONE global atomic per touched tile instead of one per (triangle, tile) pair (device-scope atomics
// are fabric transactions on this chip: 2.8 M of them cost 97 + 168 us of the 8192^2 bake).  A workgroup whose box exceeds the
// histogram (an index buffer in random order) falls back to per-pair atomics.
constexpr int kTexHist = 1024;

template <bool FILL>
__global__ __launch_bounds__(kBlock) void k_tex_bin(const TexP P)
{
    __shared__ int s_bb[4];
    __shared__ uint32_t s_hist[kTexHist], s_hbase[kTexHist];
    const int tid = threadIdx.x, lane = tid & 63;
    const int i = blockIdx.x * kBlock + tid;
    int x_min, x_max, y_min, y_max, i0 = 0, i1 = 0, i2 = 0;
    const bool on = i < P.ntri && tri_bbox(P, i, x_min, x_max, y_min, y_max, i0, i1, i2);
    const uint4 entry = make_uint4((uint32_t)i, (uint32_t)i0, (uint32_t)i1, (uint32_t)i2);
    int bx0 = 0, bx1 = -1, by0 = 0, by1 = -1;
    if (on) {
        bx0 = x_min / kTile; bx1 = x_max / kTile;
        by0 = max(y_min, P.row_begin) / kTile - P.by0; by1 = min(y_max, P.row_end - 1) / kTile - P.by0;
    }
    if (tid == 0) { s_bb[0] = 0x7fffffff; s_bb[1] = 0x7fffffff; s_bb[2] = -1; s_bb[3] = -1; }
    __syncthreads();
    {
        int a0 = on ? bx0 : 0x7fffffff, a1 = on ? by0 : 0x7fffffff, a2 = on ? bx1 : -1, a3 = on ? by1 : -1;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            a0 = min(a0, __shfl_xor(a0, d, 64)); a1 = min(a1, __shfl_xor(a1, d, 64));
            a2 = max(a2, __shfl_xor(a2, d, 64)); a3 = max(a3, __shfl_xor(a3, d, 64));
        }
        if (lane == 0) { atomicMin(&s_bb[0], a0); atomicMin(&s_bb[1], a1); atomicMax(&s_bb[2], a2); atomicMax(&s_bb[3], a3); }
    }
    __syncthreads();
    const int ox = s_bb[0], oy = s_bb[1], bw = s_bb[2] - s_bb[0] + 1, bh = s_bb[3] - s_bb[1] + 1;
    if (s_bb[2] < 0) return;                                       // nothing of this workgroup is in the band
    const int area = bw * bh;
    if (area <= kTexHist) {
        for (int e = tid; e < area; e += kBlock) s_hist[e] = 0;
        __syncthreads();
        for (int by = by0; by <= by1; by++)
            for (int bx = bx0; bx <= bx1; bx++) atomicAdd(&s_hist[(by - oy) * bw + (bx - ox)], 1u);
        __syncthreads();
        for (int e = tid; e < area; e += kBlock) {
            const uint32_t c = s_hist[e];
            if (c == 0) continue;
            const int b = (oy + e / bw) * P.bx + (ox + e % bw);
            if (FILL) { s_hbase[e] = P.bin_off[b] + atomicAdd(&P.bin_cursor[b], c); s_hist[e] = 0; }
            else atomicAdd(&P.bin_count[b], c);
        }
        if (!FILL) return;
        __syncthreads();
        for (int by = by0; by <= by1; by++)
            for (int bx = bx0; bx <= bx1; bx++) {
                const int e = (by - oy) * bw + (bx - ox);
                const uint32_t pos = s_hbase[e] + atomicAdd(&s_hist[e], 1u);
                if (pos < P.cap) P.list[pos] = entry;
            }
    } else {
        for (int by = by0; by <= by1; by++)
            for (int bx = bx0; bx <= bx1; bx++) {
                const int b = by * P.bx + bx;
                if (FILL) {
                    const uint32_t pos = P.bin_off[b] + atomicAdd(&P.bin_cursor[b], 1u);
                    if (pos < P.cap) P.list[pos] = entry;
                } else {
                    atomicAdd(&P.bin_count[b], 1u);
                }
            }
    }
}

__global__ __launch_bounds__(kScanChunk) void k_tex_chunk_sums(const TexP P)
{
    __shared__ uint32_t s_w[kScanChunk / 64];
    const int c = blockIdx.x, tid = threadIdx.x, b = c * kScanChunk + tid;
    uint32_t x = b < P.bx * P.by ? P.bin_count[b] : 0u;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) x += (uint32_t)__shfl_xor((int)x, d, 64);
    if ((tid & 63) == 0) s_w[tid >> 6] = x;
    __syncthreads();
    if (tid == 0) {
        uint32_t tot = 0;
#pragma unroll
        for (int w = 0; w < kScanChunk / 64; w++) tot += s_w[w];
        P.chunk_sum[c] = tot;
    }
}

__global__ __launch_bounds__(kScanChunk) void k_tex_scan(const TexP P)
{
    __shared__ uint32_t s_w[16];
    __shared__ unsigned long long s_carry;
    const int chunk = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nb = P.bx * P.by;
    if (wave == 0) {                                                // pairs in the chunks before this one
        unsigned long long part = 0;
        for (int i = lane; i < chunk; i += 64) part += P.chunk_sum[i];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1)
            part += ((unsigned long long)(uint32_t)__shfl_xor((int)(part >> 32), d, 64) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)part, d, 64);
        if (lane == 0) s_carry = part;
    }
    const int b = chunk * kScanChunk + tid;
    const uint32_t c = b < nb ? P.bin_count[b] : 0u;
    uint32_t incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const uint32_t x = s_w[k];
        if (k < wave) woff += x;
        tot += x;
    }
    const unsigned long long carry = s_carry;
    const unsigned long long off = carry + woff + incl - c;
    if (b < nb) P.bin_off[b] = off > 0xffffffffull ? 0xffffffffu : (uint32_t)off;
    if (tid == 0 && chunk == P.n_chunks - 1) *P.total = carry + tot;
}

// One workgroup per 32x32-texel tile.  The first version let every texel test every triangle of its tile (3.4 G tests for
// the 8192^2 bake, 1.5 of its 2.1 ms, vector-ALU bound whatever the memory side did).  Like the reference, a triangle now only
// visits the texels of its own bounding box:
//   1. set-up: one lane per triangle of the round builds its record in LDS (parallel gathers);
//   2. each wave takes records in turn and spreads the (bbox intersect tile) texels over its lanes; a texel that passes the
//      reference's test with a depth above the caller's depth buffer enters an LDS ds_max_u64 with the key
//      (order-preserving bits of its depth << 32 | ~triangle index): the maximum is the reference's final state - largest
//      depth, lowest index among equals - whatever the order of arrival;
//   3. every texel reads its winner and recomputes that one triangle's weights with the very same operations (bit-identical
//      to what the loop of step 2 saw), interpolates the colours and writes image + depth.
__device__ __forceinline__ uint32_t depth_order_bits(const float d)      // a > b  <=>  bits(a) > bits(b); -0 counts as +0
{
    const uint32_t u = __float_as_uint(d + 0.0f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// What the texel loop and the write-out read per staged triangle: 64 bytes = four 16-byte words (the set-up record `Tri` without the
// vertex indices, its box already clipped to the tile and packed as texel offsets inside it).
struct __attribute__((aligned(16))) TriRec {
    float p0x, p0y, v0x, v0y, v1x, v1y;
    float dot00, dot01, dot11, inverDeno;
    float d0, d1, d2;
    uint32_t box;                           // x_lo | x_hi << 8 | y_lo << 16 | y_hi << 24, relative to the tile; x_lo > x_hi: no texel here
    int idx;
    int pad;
};
static_assert(sizeof(TriRec) == 64, "TriRec must stay four float4 words");
constexpr uint32_t kEmptyBox = 1u;          // x_lo = 1, x_hi = 0

struct TriEval { float w0, w1, w2, pd; bool pass; };

// mesh_core.cpp:201-216 for one (triangle record, texel): barycentric weights, the in-triangle / border-ring test, depth
template <class TRI>
__device__ __forceinline__ TriEval eval_texel(const TRI &t, const float px, const float py, const bool border)
{
#pragma clang fp contract(off)
    TriEval r;
    const float v2x = px - t.p0x, v2y = py - t.p0y;
    const float dot02 = t.v0x * v2x + t.v0y * v2y;
    const float dot12 = t.v1x * v2x + t.v1y * v2y;
    const float u = (t.dot11 * dot02 - t.dot01 * dot12) * t.inverDeno;
    const float v = (t.dot00 * dot12 - t.dot01 * dot02) * t.inverDeno;
    const bool in_tri = (u >= 0) && (v >= 0) && (u + v < 1);
    r.pass = border || in_tri;
    r.w0 = 1 - u - v; r.w1 = v; r.w2 = u;
    r.pd = r.w0 * t.d0 + r.w1 * t.d1 + r.w2 * t.d2;
    return r;
}

// mesh_core.cpp:186-208: the per-triangle quantities from its three vertices (x, y, depth)
__device__ __forceinline__ void setup_from_vertices(const float x0, const float y0, const float x1, const float y1, const float x2,
                                                    const float y2, const float d0, const float d1, const float d2, Tri &t)
{
#pragma clang fp contract(off)
    t.p0x = x0; t.p0y = y0;
    t.v0x = x2 - x0; t.v0y = y2 - y0;
    t.v1x = x1 - x0; t.v1y = y1 - y0;
    t.dot00 = t.v0x * t.v0x + t.v0y * t.v0y;
    t.dot01 = t.v0x * t.v1x + t.v0y * t.v1y;
    t.dot11 = t.v1x * t.v1x + t.v1y * t.v1y;
    if (t.dot00 * t.dot11 - t.dot01 * t.dot01 == 0) t.inverDeno = 0;
    else t.inverDeno = 1 / (t.dot00 * t.dot11 - t.dot01 * t.dot01);
    t.d0 = d0; t.d1 = d1; t.d2 = d2;
}

__device__ __forceinline__ void setup_triangle(const TexP &P, const uint4 entry, Tri &t)
{
#pragma clang fp contract(off)
    const int i = (int)entry.x, i0 = (int)entry.y, i1 = (int)entry.z, i2 = (int)entry.w;
    const float x0 = P.vertices[3 * (size_t)i0], y0 = P.vertices[3 * (size_t)i0 + 1];
    const float x1 = P.vertices[3 * (size_t)i1], y1 = P.vertices[3 * (size_t)i1 + 1];
    const float x2 = P.vertices[3 * (size_t)i2], y2 = P.vertices[3 * (size_t)i2 + 1];
    t.p0x = x0; t.p0y = y0;
    t.v0x = x2 - x0; t.v0y = y2 - y0;
    t.v1x = x1 - x0; t.v1y = y1 - y0;
    t.dot00 = t.v0x * t.v0x + t.v0y * t.v0y;
    t.dot01 = t.v0x * t.v1x + t.v0y * t.v1y;
    t.dot11 = t.v1x * t.v1x + t.v1y * t.v1y;
    if (t.dot00 * t.dot11 - t.dot01 * t.dot01 == 0) t.inverDeno = 0;
    else t.inverDeno = 1 / (t.dot00 * t.dot11 - t.dot01 * t.dot01);
    t.d0 = P.vertices[3 * (size_t)i0 + 2]; t.d1 = P.vertices[3 * (size_t)i1 + 2]; t.d2 = P.vertices[3 * (size_t)i2 + 2];
    t.x_min = max((int)ceilf(fminf(x0, fminf(x1, x2))), 0);
    t.x_max = min((int)floorf(fmaxf(x0, fmaxf(x1, x2))), P.w - 1);
    t.y_min = max((int)ceilf(fminf(y0, fminf(y1, y2))), 0);
    t.y_max = min((int)floorf(fmaxf(y0, fmaxf(y1, y2))), P.h - 1);
    t.idx = i; t.i0 = i0; t.i1 = i1;      // (i2 = entry.w)
}

// FRESH: the call starts from render.py:72's state - image = background (zeros), depth buffer = -999999 everywhere - which is
// what `render_colors` always does: the kernel then neither reads the depth buffer nor needs the caller to fill 2 x h x w x 4
// bytes first; it writes EVERY texel of the band (winner or background).
// Tiles whose list fits one staging round (<= kStage triangles: every tile of a UV mesh baked at the usual 8 texels per edge)
// take the fast path: the records stay in LDS together with their vertex colours, and the key carries the record's staged SLOT
// below the (inverted) triangle index, so that the write-out reads the winner's record from LDS instead of gathering triangle ->
// vertices -> colours per texel (1.0 of the 1.48 ms of the 8192^2 bake in round 2).  (Until round 5 the slot WAS the order: the
// list was rank-sorted by index first - two barriers and an n^2 loop per tile, 0.018 ms of the bake.)
constexpr float kFreshDepth = -999999.0f;
constexpr int kSlotBits = 7, kIdxBits = 32 - kSlotBits;   // fast path: key low word = ~index (25 bits) | staged slot (7 bits)
static_assert(kStage <= (1 << kSlotBits), "a staged slot must fit its bits of the key");
constexpr int kMaxC = 3;                                 // colour channels the fast path stages (Topo4D: 3; more take the general path)

template <bool FRESH>
// Eight waves per SIMD (at most 64 registers; it took 78 and ran six): the tile is a chain of dependent fetches - bin header, list
// (which carries the vertex indices: the fill pass has them in registers anyway), vertices | colours - and what hides it is other workgroups.  8192^2 bake on one box: 0.618 ms at six, 0.590 at seven, 0.563 at eight.
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(FRESH ? 8 : 6, FRESH ? 8 : 6))) void k_tex_render(const TexP P)
{
#pragma clang fp contract(off)
    __shared__ TriRec s_tri[kStage];
    __shared__ float s_col[kStage][3][kMaxC];
    __shared__ unsigned long long s_key[kTile * kTile];
    __shared__ float s_depth[FRESH ? 1 : kTile * kTile];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b = blockIdx.x;
    const int bxi = b % P.bx, byi = b / P.bx + P.by0;
    const uint32_t n = P.bin_count[b];
    const int tx0 = bxi * kTile, ty0 = byi * kTile;
    // rows of the tile that belong to this call's band and to the image
    const int ry_lo = max(ty0, P.row_begin), ry_hi = min(min(ty0 + kTile, P.h), P.row_end) - 1, rx_hi = min(tx0 + kTile, P.w) - 1;
    constexpr int kPer = kTile * kTile / kBlock;
    if (n == 0) {
        if (FRESH) {                                               // nobody draws here: background and the initial depth
#pragma unroll
            for (int j = 0; j < kPer; j++) {
                const int e = tid + j * kBlock;
                const int x = tx0 + (e & (kTile - 1)), y = ty0 + e / kTile;
                if (x > rx_hi || y < ry_lo || y > ry_hi) continue;
                const size_t o = (size_t)y * P.w + x;
                for (int k = 0; k < P.c; k++) P.image[o * P.c + k] = P.bg ? P.bg[o * P.c + k] : 0.f;
                P.depth[o] = kFreshDepth;
            }
        }
        return;
    }
    const uint32_t off = P.bin_off[b];
    // (only the tiles along the image's edge hold texels of the 2-pixel border ring: the others skip the four compares per texel)
    const bool tile_border = tx0 < 2 || ty0 < 2 || tx0 + kTile > P.w - 3 || ty0 + kTile > P.h - 3;
    // the staged record of triangle record `t`: its box clipped to this tile (and band) once, here, instead of in every wave that takes it
    auto make_rec = [&](const Tri &t) {
        TriRec r;
        r.p0x = t.p0x; r.p0y = t.p0y; r.v0x = t.v0x; r.v0y = t.v0y; r.v1x = t.v1x; r.v1y = t.v1y;
        r.dot00 = t.dot00; r.dot01 = t.dot01; r.dot11 = t.dot11; r.inverDeno = t.inverDeno;
        r.d0 = t.d0; r.d1 = t.d1; r.d2 = t.d2;
        const int x_lo = max(t.x_min, tx0), x_hi = min(t.x_max, rx_hi), y_lo = max(t.y_min, ry_lo), y_hi = min(t.y_max, ry_hi);
        r.box = (x_hi < x_lo || y_hi < y_lo) ? kEmptyBox
                                             : (uint32_t)(x_lo - tx0) | (uint32_t)(x_hi - tx0) << 8 | (uint32_t)(y_lo - ty0) << 16 | (uint32_t)(y_hi - ty0) << 24;
        r.idx = t.idx; r.pad = 0;
        return r;
    };
    // One staging round + triangle indices that leave seven bits of the key's low word for the staged slot (below)
    const bool fast = n <= (uint32_t)kStage && P.c <= kMaxC && off + n <= P.cap && P.ntri <= (1 << kIdxBits);
    if (fast && tid < (int)n) {
        // ---- records + vertex colours of every triangle of the tile (in the list's order of arrival)
        const uint4 entry = P.list[off + tid];
        Tri t;
        setup_triangle(P, entry, t);
        s_tri[tid] = make_rec(t);
        for (int k = 0; k < P.c; k++) {
            s_col[tid][0][k] = P.colors[(size_t)P.c * entry.y + k];
            s_col[tid][1][k] = P.colors[(size_t)P.c * entry.z + k];
            s_col[tid][2][k] = P.colors[(size_t)P.c * entry.w + k];
        }
    }
    for (int e = tid; e < kTile * kTile; e += kBlock) {           // the caller's depth buffer for this tile: read once, tested from LDS
        s_key[e] = 0ull;
        if (!FRESH) {
            const int x = tx0 + (e & (kTile - 1)), y = ty0 + e / kTile;
            s_depth[e] = (x <= rx_hi && y >= ry_lo && y <= ry_hi) ? P.depth[(size_t)y * P.w + x] : 0.f;
        }
    }
    __syncthreads();
    for (uint32_t base = 0; base < n; base += kStage) {
        const int cnt = (int)min((uint32_t)kStage, n - base);
        if (!fast) {
            __syncthreads();
            if (tid < cnt) {
                Tri t;
                if (off + base + tid < P.cap) setup_triangle(P, P.list[off + base + tid], t);
                else { memset(&t, 0, sizeof(t)); t.x_min = 1; t.x_max = 0; t.y_min = 1; t.y_max = 0; t.idx = 0x7fffffff; }     // matches no texel
                s_tri[tid] = make_rec(t);
            }
            __syncthreads();
        }
        // FOUR records per wave at a time, sixteen lanes each: a triangle of a UV mesh baked at 8 texels per edge has a clipped box of
        // 64-81 texels, so a whole wave per record spent its second pass on 0-17 of 64 lanes (8192^2: 2.1 M triangles x 128 lane
        // slots for 170 M box texels); sixteen lanes per record take the same box in 4-6 passes of 16 (64-96 slots).  Which lane
        // evaluates a texel does not matter: the LDS maximum is order-independent.
        constexpr int kG = T4D_TEX_GROUP, kPerWave = 64 / kG;             // lanes per record, records per wave at a time
        const int grp = lane / kG, hl = lane % kG;
        for (int k0 = kPerWave * wave; k0 < cnt; k0 += kPerWave * (kBlock / 64)) {
            const int k = k0 + grp;
            const bool have = k < cnt;
            const TriRec t = s_tri[have ? k : k0];
            const int bx_lo = (int)(t.box & 0xffu), bx_hi = (int)((t.box >> 8) & 0xffu), by_lo = (int)((t.box >> 16) & 0xffu), by_hi = (int)(t.box >> 24);
            const int rw = bx_hi - bx_lo + 1, rh = by_hi - by_lo + 1;
            const int npx = (have && rw > 0) ? rw * rh : 0;
            int nmax = 0;                                                                                     // wave-uniform trip count
#pragma unroll
            for (int g = 0; g < kPerWave; g++) nmax = max(nmax, __builtin_amdgcn_readlane(npx, g * kG));
            // the key's low word: the lower triangle index wins on equal depth (mesh_core.cpp:213 `>` keeps the first); the fast path
            // appends the staged slot so that the write-out finds the winner's record in LDS without a search
            const uint32_t low = fast ? ((~(uint32_t)t.idx & ((1u << kIdxBits) - 1u)) << kSlotBits | (uint32_t)k) : ~(uint32_t)t.idx;
            // p / rw without an integer division (~25 instructions per texel, a fifth of this loop): rw <= 32 and p < 1,024, so the
            // 16-bit fixed-point reciprocal m >= 65536 / rw with m rw - 65536 <= rw gives the exact quotient (p (m rw - 65536) < 65536)
            const uint32_t m_rw = (uint32_t)(65536.0f * __builtin_amdgcn_rcpf((float)max(rw, 1))) + 1u;
            for (int p = hl; p < nmax; p += kG) {
                if (p >= npx) continue;
                const int dy = (int)(((uint32_t)p * m_rw) >> 16), lx = bx_lo + (p - dy * rw), ly = by_lo + dy;     // inside the tile
                const float px = (float)(tx0 + lx), py = (float)(ty0 + ly);
                const bool border = tile_border && (px < 2 || px > P.w - 3 || py < 2 || py > P.h - 3);      // mesh_core.cpp:211
                const TriEval ev = eval_texel(t, px, py, border);
                // `pd > depth_buffer` against the caller's buffer first (also drops NaN); later rivals meet in the LDS maximum
                const float have_d = FRESH ? kFreshDepth : s_depth[ly * kTile + lx];
                if (ev.pass && ev.pd > have_d)
                    atomicMax(&s_key[ly * kTile + lx], ((unsigned long long)depth_order_bits(ev.pd) << 32) | low);
            }
        }
    }
    __syncthreads();
    // The image is [h, w, c]: a texel's c floats sit 4c bytes from its neighbour's; consecutive lanes therefore take
    // consecutive texels of a tile row (the stores of a wave cover whole cache lines between them).
    if (fast) {
        // the winner's record and colours are still in LDS: same record, same operations as in the loop above.
        // (Measured and dropped in round 5: staging a whole FRESH tile's colours in 12 KB of LDS and writing them as 16-byte stores
        // instead of three 4-byte stores per lane at a 12-byte stride - 0.622 -> 0.754 ms for the 8192^2 bake: the 37 KB of LDS
        // leave four workgroups per CU where six ran, and the tile pays one more barrier.)
#pragma unroll
        for (int j = 0; j < kPer; j++) {
            const int e = tid + j * kBlock;
            const int x = tx0 + (e & (kTile - 1)), y = ty0 + e / kTile;
            if (x > rx_hi || y < ry_lo || y > ry_hi) continue;
            const unsigned long long key = s_key[e];
            const size_t o = (size_t)y * P.w + x;
            if (key == 0ull) {                                     // nobody drew this texel
                if (FRESH) {
                    for (int k = 0; k < P.c; k++) P.image[o * P.c + k] = P.bg ? P.bg[o * P.c + k] : 0.f;
                    P.depth[o] = kFreshDepth;
                }
                continue;
            }
            const int slot = (int)((uint32_t)key & ((1u << kSlotBits) - 1u));
            const float px = (float)x, py = (float)y;
            const bool border = tile_border && (px < 2 || px > P.w - 3 || py < 2 || py > P.h - 3);
            const TriEval ev = eval_texel(s_tri[slot], px, py, border);
            for (int k = 0; k < P.c; k++)
                P.image[o * P.c + k] = ev.w0 * s_col[slot][0][k] + ev.w1 * s_col[slot][1][k] + ev.w2 * s_col[slot][2][k];
            P.depth[o] = ev.pd;
        }
        return;
    }
    // Four texels per thread in two halves of two, their dependent gathers (triangle -> vertices -> colours) issued level by level for
    // both texels of a half: the chain's latency is paid twice per workgroup, not once per texel (all four at once need 52 registers
    // for the gathered values alone: at this kernel's 64 they spilled, 8.4 M triangles 1.13 -> 1.32 ms).
    constexpr int kHalf = kPer / 2;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        int wi[kHalf], wv[kHalf][3];
        bool hit[kHalf];
#pragma unroll
        for (int j = 0; j < kHalf; j++) {
            const unsigned long long key = s_key[tid + (h * kHalf + j) * kBlock];
            hit[j] = key != 0ull;                                  // nobody drew this texel: the caller's background stays
            wi[j] = hit[j] ? (int)~(uint32_t)key : 0;
        }
#pragma unroll
        for (int j = 0; j < kHalf; j++)
#pragma unroll
            for (int m = 0; m < 3; m++) wv[j][m] = hit[j] ? P.triangles[3 * (size_t)wi[j] + m] : 0;
        float vx[kHalf][3], vy[kHalf][3], vz[kHalf][3];
#pragma unroll
        for (int j = 0; j < kHalf; j++)
#pragma unroll
            for (int m = 0; m < 3; m++) {
                vx[j][m] = hit[j] ? P.vertices[3 * (size_t)wv[j][m]] : 0.f;
                vy[j][m] = hit[j] ? P.vertices[3 * (size_t)wv[j][m] + 1] : 0.f;
                vz[j][m] = hit[j] ? P.vertices[3 * (size_t)wv[j][m] + 2] : 0.f;
            }
#pragma unroll
        for (int j = 0; j < kHalf; j++) {
            const int e = tid + (h * kHalf + j) * kBlock;
            const int x = tx0 + (e & (kTile - 1)), y = ty0 + e / kTile;
            if (!hit[j]) {
                if (FRESH && x <= rx_hi && y >= ry_lo && y <= ry_hi) {
                    const size_t o = (size_t)y * P.w + x;
                    for (int k = 0; k < P.c; k++) P.image[o * P.c + k] = P.bg ? P.bg[o * P.c + k] : 0.f;
                    P.depth[o] = kFreshDepth;
                }
                continue;
            }
            Tri t;
            setup_from_vertices(vx[j][0], vy[j][0], vx[j][1], vy[j][1], vx[j][2], vy[j][2], vz[j][0], vz[j][1], vz[j][2], t);
            const float px = (float)x, py = (float)y;
            const bool border = px < 2 || px > P.w - 3 || py < 2 || py > P.h - 3;
            const TriEval ev = eval_texel(t, px, py, border);
            float *out = P.image + ((size_t)y * P.w + x) * P.c;
            for (int k = 0; k < P.c; k++)
                out[k] = ev.w0 * P.colors[(size_t)P.c * wv[j][0] + k] + ev.w1 * P.colors[(size_t)P.c * wv[j][1] + k] +
                         ev.w2 * P.colors[(size_t)P.c * wv[j][2] + k];
            P.depth[(size_t)y * P.w + x] = ev.pd;
        }
    }
}

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

struct TexLayout { size_t total_, bin_count, bin_cursor, zero_end, bin_off, chunk_sum, list, bytes; };

TexLayout tex_layout(int h, int w, int64_t cap)
{
    const size_t nb = (size_t)((w + kTile - 1) / kTile) * ((h + kTile - 1) / kTile);
    TexLayout L;
    size_t o = 0;
    L.total_ = o;     o = align_up(o + 8);
    L.bin_count = o;  o = align_up(o + nb * 4);
    L.bin_cursor = o; o = align_up(o + nb * 4);
    L.zero_end = o;
    L.bin_off = o;    o = align_up(o + nb * 4);
    L.chunk_sum = o;  o = align_up(o + ((nb + kScanChunk - 1) / kScanChunk) * 4);
    L.list = o;       o = align_up(o + (size_t)cap * sizeof(uint4));
    L.bytes = o;
    return L;
}

}  // namespace

T4D_EXPORT size_t t4d_texture_bake_scratch_bytes(int32_t h, int32_t w, int64_t pair_capacity)
{
    if (h < 1 || w < 1 || pair_capacity < 1) return 0;
    return tex_layout(h, w, pair_capacity).bytes;
}

namespace {
int texture_bake_impl(const bool fresh, const float *bg, const float *vertices, const int32_t *triangles, const float *colors, int32_t nver,
                      int32_t ntri, int32_t h, int32_t w, int32_t c, int32_t row_begin, int32_t row_end, float *image, float *depth_buffer,
                      void *scratch, size_t scratch_bytes, int64_t pair_capacity, int64_t *pairs_needed, void *hip_stream)
{
    if (!vertices || !triangles || !colors || !image || !depth_buffer || !scratch || nver < 1 || ntri < 0 || h < 1 || w < 1 || c < 1)
        return t4d_internal_fail(T4D_ERR_ARG, "t4d_texture_bake: bad arguments%s", "");
    if (row_begin < 0 || row_end > h || row_begin >= row_end)
        return t4d_internal_fail(T4D_ERR_ARG, "t4d_texture_bake: row band must satisfy 0 <= begin < end <= h%s", "");
    if (pair_capacity < 1 || pair_capacity > 0x7fffffffLL)
        return t4d_internal_fail(T4D_ERR_ARG, "t4d_texture_bake: pair_capacity out of range%s", "");
    const TexLayout L = tex_layout(h, w, pair_capacity);
    if (scratch_bytes < L.bytes) return t4d_internal_fail(T4D_ERR_STATE_SIZE, "t4d_texture_bake: scratch too small%s", "");
    hipStream_t stream = (hipStream_t)hip_stream;
    char *sc = (char *)scratch;
    TexP P;
    memset(&P, 0, sizeof(P));
    P.vertices = vertices; P.triangles = triangles; P.colors = colors;
    P.nver = nver; P.ntri = ntri; P.h = h; P.w = w; P.c = c; P.row_begin = row_begin; P.row_end = row_end;
    P.bx = (w + kTile - 1) / kTile;
    P.by0 = row_begin / kTile;
    P.by = (row_end - 1) / kTile - P.by0 + 1;
    P.cap = (uint32_t)pair_capacity;
    P.total = (unsigned long long *)(sc + L.total_);
    P.bin_count = (uint32_t *)(sc + L.bin_count);
    P.bin_cursor = (uint32_t *)(sc + L.bin_cursor);
    P.bin_off = (uint32_t *)(sc + L.bin_off);
    P.chunk_sum = (uint32_t *)(sc + L.chunk_sum);
    P.n_chunks = (P.bx * P.by + kScanChunk - 1) / kScanChunk;
    P.list = (uint4 *)(sc + L.list);
    P.image = image; P.depth = depth_buffer; P.bg = bg;
    if (pairs_needed) *pairs_needed = 0;
#define TEX_HIP(call)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) return t4d_internal_fail(T4D_ERR_HIP, #call ": %s", hipGetErrorString(e_)); \
    } while (0)
    TEX_HIP(hipMemsetAsync(sc, 0, L.zero_end, stream));
    unsigned long long total = 0;
    if (ntri > 0) {
        const int gt = (ntri + kBlock - 1) / kBlock;
        hipLaunchKernelGGL(k_tex_bin<false>, dim3(gt), dim3(kBlock), 0, stream, P);
        if (P.n_chunks > 1) hipLaunchKernelGGL(k_tex_chunk_sums, dim3(P.n_chunks), dim3(kScanChunk), 0, stream, P);
        hipLaunchKernelGGL(k_tex_scan, dim3(P.n_chunks), dim3(kScanChunk), 0, stream, P);
        TEX_HIP(hipMemcpyAsync(&total, P.total, 8, hipMemcpyDeviceToHost, stream));
        TEX_HIP(hipStreamSynchronize(stream));          // once per bake (a per-frame export step, not the training loop)
        if (pairs_needed) *pairs_needed = (int64_t)total;
        if (total > (unsigned long long)pair_capacity)
            return t4d_internal_fail(T4D_ERR_PAIR_OVERFLOW, "t4d_texture_bake: pair_capacity too small%s", "");
        hipLaunchKernelGGL(k_tex_bin<true>, dim3(gt), dim3(kBlock), 0, stream, P);
    } else if (!fresh) {
        return T4D_OK;
    }
    if (fresh) hipLaunchKernelGGL(k_tex_render<true>, dim3(P.bx * P.by), dim3(kBlock), 0, stream, P);
    else hipLaunchKernelGGL(k_tex_render<false>, dim3(P.bx * P.by), dim3(kBlock), 0, stream, P);
    TEX_HIP(hipGetLastError());
#undef TEX_HIP
    return T4D_OK;
}
}  // namespace

T4D_EXPORT int t4d_texture_bake(const float *vertices, const int32_t *triangles, const float *colors, int32_t nver, int32_t ntri,
                                int32_t h, int32_t w, int32_t c, int32_t row_begin, int32_t row_end, float *image,
                                float *depth_buffer, void *scratch, size_t scratch_bytes, int64_t pair_capacity,
                                int64_t *pairs_needed, void *hip_stream)
{
    return texture_bake_impl(false, nullptr, vertices, triangles, colors, nver, ntri, h, w, c, row_begin, row_end, image, depth_buffer,
                             scratch, scratch_bytes, pair_capacity, pairs_needed, hip_stream);
}

T4D_EXPORT int t4d_texture_render_colors(const float *vertices, const int32_t *triangles, const float *colors, const float *background,
                                         int32_t nver, int32_t ntri, int32_t h, int32_t w, int32_t c, int32_t row_begin,
                                         int32_t row_end, float *image, float *depth_buffer, void *scratch, size_t scratch_bytes,
                                         int64_t pair_capacity, int64_t *pairs_needed, void *hip_stream)
{
    return texture_bake_impl(true, background, vertices, triangles, colors, nver, ntri, h, w, c, row_begin, row_end, image, depth_buffer,
                             scratch, scratch_bytes, pair_capacity, pairs_needed, hip_stream);
}
