// t4d_photometric.hip — fused photometric loss (forward + gradient) for Topo4D's render loop on MI355X
// (SURVEY.md §8 row a13 / §8f rank 1).
//
// Restates reference train.py:310,315 with helpers.py:115-116 (`l1_loss_v1`) and external.py:73-116 (`calc_ssim`):
//     im' = exp(cam_m[c]) * im + cam_c[c]
//     loss_v = 0.8 * mean|im' - gt| + 0.2 * (1 - mean SSIM_11x11(im', gt))          (mean over the 3*H*W values of view v)
// SSIM: depthwise 11x11 Gaussian window (sigma 1.5), zero padding, c1 = 0.01^2, c2 = 0.03^2.
// The reference spends 5 conv2d launches forward and their autograd backward per iteration; here one launch set per
// batch of V views produces the per-view losses AND dL/d(im) (the rasterizer's backward input) AND dL/d(cam_m, cam_c):
//   k_photo_stats   per (view, channel, 16x16 tile): stage im', gt with a 5-px halo in LDS, separable 11-tap filter of
//                   (x, y, x^2, y^2, xy), SSIM map + L1 term -> partial loss sums, and the three adjoint maps
//                   D1 = g*dS/dmu1, D2 = g*dS/dE[x^2], D3 = g*dS/dE[xy]
//   k_photo_grad    per tile: separable filter of D1, D2, D3 (the window is symmetric: adjoint = same filter),
//                   dL/dx' = G*D1 + 2x'(G*D2) + y(G*D3) + L1 term; affine backward; partial sums for cam_m / cam_c
//   k_photo_final   fixed-order sums of the partials (deterministic)
// Pinned by tests against topo4d_amd/loss.py, itself pinned by golden G3 captured from the real reference functions.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>

#include "../../include/topo4d_raster.h"

#define T4D_EXPORT extern "C" __attribute__((visibility("default")))
int t4d_internal_fail(int code, const char *fmt, const char *a);

namespace {

// One kernel, no intermediate in memory.  (Rounds 1-2 ran two tile kernels with the three adjoint maps going through HBM in
// between: 302 MB written and - with the halo - 570 MB read back per 24 x 512^2 views, 0.46 ms of kernels against a traffic
// floor of ~0.1 ms.)  A workgroup of 128 threads owns a vertical STRIP of one channel of one view - up to 118 output columns, a
// segment of rows - and streams down it one image row per iteration; every thread owns ONE column:
//   * the row's (x', gt) go to LDS; each thread takes the 11-tap HORIZONTAL sums of (x, y, x^2, y^2, xy) at its column and
//     pushes them into a window of the last eleven rows that lives in REGISTERS; the VERTICAL sums over that window are the
//     five filtered maps at the row five behind - SSIM, loss terms and the three adjoint values D1..D3 of that pixel;
//   * the adjoint row goes to LDS (its columns reach five beyond the strip on either side, which is why a strip of 118 uses
//     128 threads); each thread takes its horizontal 11-tap sums, pushes them into a second register window, and the
//     vertical sums over it give G*D1, G*D2, G*D3 at the row five further behind: dL/dx' and the affine backward.
// Both 2-D windows are separable and symmetric, so adjoint = same filter.  Rows are unrolled eleven at a time so that the
// window slots are compile-time register names (no shifting).  One barrier per row; the only redundancy is the warm-up of a
// segment (20 rows) and the 10 halo columns of a strip.
constexpr int kR = 5;              // window radius (11 taps)
// threads per workgroup = columns of the first stage = output columns of a strip + 10: the launch picks the instantiation
// (64, 128, 192 or 256 threads) that covers the image width with the fewest thread-columns (512 wide: 3 strips of 171 columns
// on 192 threads; 375 wide: 7 strips of 54 on 64 threads)
constexpr int kBlock = 256;        // (k_photo_final and the masked-L1 kernels)
constexpr float kC1 = 0.01f * 0.01f, kC2 = 0.03f * 0.03f;

struct PhP {
    int V, H, W, tx, ty;           // tx strips of tw columns, ty segments of th rows per (view, channel)
    int tw, th;
    const float *im, *gt, *cam_m, *cam_c, *weight;
    float *loss, *dL_dim, *dL_dm, *dL_dc;
    float *part_loss;    // [V*3*tiles][2]  (sum |x'-y|, sum S)
    float *part_cam;     // [V*3*tiles][2]  (sum g'*(x'-c), sum g')
    float win[11];
};

__device__ __forceinline__ float block_sum(float v, float *s_red)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    const int tid = threadIdx.x, nw = blockDim.x >> 6;
    __syncthreads();
    if ((tid & 63) == 0) s_red[tid >> 6] = v;
    __syncthreads();
    float t = s_red[0];
    for (int w = 1; w < nw; w++) t += s_red[w];
    return t;
}

template <int N> struct IC { static constexpr int value = N; };
typedef float v2f __attribute__((ext_vector_type(2)));      // packed-math pair: one v_pk_fma_f32 does two of the filter's multiply-adds

#ifndef T4D_PH_WAVES
#define T4D_PH_WAVES 3             // 166 registers: three waves per SIMD (four = 128 registers spills in the row loop: 258 -> 427 us)
#endif
template <int kFT>
__global__ __launch_bounds__(kFT) __attribute__((amdgpu_waves_per_eu(T4D_PH_WAVES, T4D_PH_WAVES))) void k_photo_fused(const PhP P)
{
    constexpr int kInW = kFT + 2 * kR;                   // input columns a row needs: 138
    __shared__ float2 s_in[2][kInW];                     // (x', gt) of the current row, zero padded
    // adjoint row (D1, D2 | D3) at the first stage's columns (+ slack: the last ten threads of the second stage read beyond them
    // and emit nothing)
    __shared__ float2 s_d01[2][kFT + 2 * kR + 2];
    __shared__ float s_d2[2][kFT + 2 * kR + 2];
    __shared__ float s_red[4];
    const int tid = threadIdx.x;
    const int vc = blockIdx.z, v = vc / 3;
    const int xs = blockIdx.x * P.tw, xe = min(xs + P.tw, P.W);              // output columns [xs, xe)
    const int y0 = blockIdx.y * P.th, y1 = min(y0 + P.th, P.H);              // output rows [y0, y1)
    const size_t HW = (size_t)P.H * P.W;
    const float *im = P.im + (size_t)vc * HW, *gt = P.gt + (size_t)vc * HW;
    const float em = P.cam_m ? expf(P.cam_m[vc]) : 1.f, cc = P.cam_c ? P.cam_c[vc] : 0.f;
    const float N = 3.f * (float)HW, wv = P.weight ? P.weight[v] : 1.f;
    const float g = -0.2f * wv / N;                      // dL/dS
    const float l1w = 0.8f * wv / N;
    float w[11];
#pragma unroll
    for (int k = 0; k < 11; k++) w[k] = P.win[k];

    const int gx1 = xs - kR + tid;                       // this thread's column in the first stage
    const bool col1 = gx1 >= 0 && gx1 < P.W;
    const bool own1 = tid >= kR && gx1 < xe;             // ... which belongs to this strip's outputs (loss terms are counted once)
    const int gx2 = xs + tid;                            // ... and in the second stage
    const bool col2 = gx2 < xe;
    // columns this thread loads of every input row: xs - 10 + tid and (the first ten threads) 128 further right
    const int lx0 = xs - 2 * kR + tid, lx1 = lx0 + kFT;
    const bool l0 = lx0 >= 0 && lx0 < P.W, l1 = tid < 2 * kR && lx1 < P.W;

    // the two register windows (slots are compile-time indices); pairs of maps share a 64-bit register pair so that the
    // vertical sums run as packed multiply-adds: (x, y), (x^2, y^2) | xy  and  (D1, D2) | D3
    v2f h01[11], h23[11], hd01[11];
    float h4[11], hd2[11];
#pragma unroll
    for (int k = 0; k < 11; k++) {
        h01[k] = h23[k] = hd01[k] = (v2f){ 0.f, 0.f };
        h4[k] = hd2[k] = 0.f;
    }
    for (int b = 0; b < 2; b++) {                        // the first iteration's second stage reads a row nobody wrote
        s_d01[b][tid] = make_float2(0.f, 0.f); s_d2[b][tid] = 0.f;
        if (tid < 2 * kR + 2) { s_d01[b][kFT + tid] = make_float2(0.f, 0.f); s_d2[b][kFT + tid] = 0.f; }
    }
    float sum_l1 = 0.f, sum_s = 0.f, sum_gm = 0.f, sum_gc = 0.f;
    const int i_first = y0 - 2 * kR, i_last = y1 + 2 * kR;      // input rows i_first .. i_last (the last one only drains)

    // prefetch of the first row
    float pa0 = 0.f, pb0 = 0.f, pa1 = 0.f, pb1 = 0.f;
    auto fetch = [&](int i) {
        pa0 = pb0 = pa1 = pb1 = 0.f;
        if (i >= 0 && i < P.H) {
            const size_t o = (size_t)i * P.W;
            if (l0) { pa0 = im[o + lx0]; pb0 = gt[o + lx0]; }
            if (l1) { pa1 = im[o + lx1]; pb1 = gt[o + lx1]; }
        }
    };
    fetch(i_first);

    auto row = [&](const int i, auto J_) {
        constexpr int J = decltype(J_)::value;           // slot of this row in both windows
        const int buf = i & 1;
        // ---- this row's inputs to LDS (zero padding outside the image: external.py:86 padding=5), next row's loads in flight
        {
            const bool in_img = i >= 0 && i < P.H;
            s_in[buf][tid] = (in_img && l0) ? make_float2(em * pa0 + cc, pb0) : make_float2(0.f, 0.f);
            if (tid < 2 * kR) s_in[buf][tid + kFT] = (in_img && l1) ? make_float2(em * pa1 + cc, pb1) : make_float2(0.f, 0.f);
        }
        fetch(i + 1);
        // the pixel of the OUTPUT row of this iteration (eleven rows behind): needed at the very end, requested now
        const int o_row = i - 2 * kR - 1;
        const bool emit = col2 && o_row >= y0 && o_row < y1;
        float o_im = 0.f, o_gt = 0.f;
        if (emit) { o_im = im[(size_t)o_row * P.W + gx2]; o_gt = gt[(size_t)o_row * P.W + gx2]; }
#ifndef T4D_PH_NOBARRIER          // (timing experiment: what the one barrier per row costs; results are wrong without it)
        __syncthreads();
#endif
        // ---- first stage, horizontal: 11 taps of (x, y, x^2, y^2, xy) at this thread's column
        {
            v2f a01 = { 0.f, 0.f }, a23 = { 0.f, 0.f };
            float a4 = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) {
                const v2f ab = *reinterpret_cast<const v2f *>(&s_in[buf][tid + k]);
                const v2f wab = w[k] * ab;                       // four instructions per tap for the five sums
                a01 += wab;
                a23 = __builtin_elementwise_fma(wab, ab, a23);
                a4 = fmaf(wab.x, ab.y, a4);
            }
            h01[J] = a01; h23[J] = a23; h4[J] = a4;
        }
        // ---- first stage, vertical: rows i-10 .. i are in slots J+1 .. J+11 (mod 11) -> the filtered maps at row s = i - 5
        {
            v2f m12 = { 0.f, 0.f }, eac = { 0.f, 0.f };
            float eb = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) {
                const int sl = (J + 1 + k) % 11;
                const v2f wk = { w[k], w[k] };
                m12 = __builtin_elementwise_fma(wk, h01[sl], m12);
                eac = __builtin_elementwise_fma(wk, h23[sl], eac);
                eb = fmaf(w[k], h4[sl], eb);
            }
            const float mu1 = m12.x, mu2 = m12.y, ea = eac.x, ec = eac.y;
            const int srow = i - kR;
            float d1 = 0.f, d2 = 0.f, d3 = 0.f;
            if (col1 && srow >= 0 && srow < P.H) {               // the adjoint map exists inside the image only
                const float mu1s = mu1 * mu1, mu2s = mu2 * mu2, mu12 = mu1 * mu2;
                const float s11 = ea - mu1s, s22 = ec - mu2s, s12 = eb - mu12;
                const float A1 = 2.f * mu12 + kC1, A2 = 2.f * s12 + kC2, B1 = mu1s + mu2s + kC1, B2 = s11 + s22 + kC2;
                const float inv = 1.f / (B1 * B2);               // the one division of the pixel: 1/B1 = B2 * inv, 1/B2 = B1 * inv
                const float S = A1 * A2 * inv;
                // S = A1 A2 / (B1 B2) with s11 = a - mu1^2, s12 = b - mu1 mu2 (a, b, c = filtered x^2, xy, y^2)
                const float gi = g * inv;
                d1 = 2.f * gi * (mu2 * (A2 - A1) - S * mu1 * (B2 - B1));
                d2 = -gi * S * B1;
                d3 = 2.f * gi * A1;
                if (own1 && srow >= y0 && srow < y1) sum_s += S;
            }
            s_d01[buf][tid] = make_float2(d1, d2); s_d2[buf][tid] = d3;
        }
        // ---- second stage on the adjoint row written ONE iteration ago (made visible by this iteration's barrier)
        {
            const int pb = buf ^ 1;
            v2f q01 = { 0.f, 0.f };
            float q2 = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) {
                const v2f wk = { w[k], w[k] };
                q01 = __builtin_elementwise_fma(wk, *reinterpret_cast<const v2f *>(&s_d01[pb][tid + k]), q01);
                q2 = fmaf(w[k], s_d2[pb][tid + k], q2);
            }
            constexpr int J2 = (J + 10) % 11;            // the adjoint row of the previous iteration sits one slot back
            hd01[J2] = q01; hd2[J2] = q2;
            v2f r01 = { 0.f, 0.f };
            float r2 = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) {
                const int sl = (J2 + 1 + k) % 11;
                const v2f wk = { w[k], w[k] };
                r01 = __builtin_elementwise_fma(wk, hd01[sl], r01);
                r2 = fmaf(w[k], hd2[sl], r2);
            }
            const float r0 = r01.x, r1 = r01.y;
            if (emit) {
                const float x = em * o_im + cc, y = o_gt;
                const float d = x - y;
                sum_l1 += fabsf(d);
                const float gl1 = l1w * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
                const float gg = r0 + 2.f * x * r1 + y * r2 + gl1;                   // dL/dx'
                P.dL_dim[(size_t)vc * HW + (size_t)o_row * P.W + gx2] = em * gg;
                sum_gm += gg * (em * o_im);                                          // d x'/d cam_m = exp(cam_m) * im
                sum_gc += gg;
            }
        }
    };
    for (int base = i_first; base <= i_last; base += 11) {
        if (base + 0 <= i_last) row(base + 0, IC<0>()); if (base + 1 <= i_last) row(base + 1, IC<1>());
        if (base + 2 <= i_last) row(base + 2, IC<2>()); if (base + 3 <= i_last) row(base + 3, IC<3>());
        if (base + 4 <= i_last) row(base + 4, IC<4>()); if (base + 5 <= i_last) row(base + 5, IC<5>());
        if (base + 6 <= i_last) row(base + 6, IC<6>()); if (base + 7 <= i_last) row(base + 7, IC<7>());
        if (base + 8 <= i_last) row(base + 8, IC<8>()); if (base + 9 <= i_last) row(base + 9, IC<9>());
        if (base + 10 <= i_last) row(base + 10, IC<10>());
    }
    const float tl1 = block_sum(sum_l1, s_red), tss = block_sum(sum_s, s_red);
    const float tgm = block_sum(sum_gm, s_red), tgc = block_sum(sum_gc, s_red);
    if (tid == 0) {
        const size_t t = ((size_t)vc * P.ty + blockIdx.y) * P.tx + blockIdx.x;
        P.part_loss[2 * t] = tl1; P.part_loss[2 * t + 1] = tss;
        P.part_cam[2 * t] = tgm; P.part_cam[2 * t + 1] = tgc;
    }
}

// The same strip, two WAVE ROLES (round 3): threads [0, kFT) run the first stage - inputs, the five filtered maps, SSIM, the
// adjoint row - and threads [kFT, 2 kFT) the second - the three filtered adjoint maps, dL/dx', the affine backward.  One
// thread per column held both register windows (55 + 33 values) and their arithmetic: 166 registers, three waves per SIMD,
// and at 24 x 512^2 only 2,592 waves for 1,024 SIMDs - the vector ALUs were busy 57 % of the kernel (rocprofv3 PMC: 78 M
// instructions = 128 us of issue in a 237 us kernel; the rest is a wave waiting for its own LDS round trips and barrier).
// Split by stage a thread needs ~100 / ~70 registers and there are twice as many waves.  The arithmetic, its order and the
// one barrier per row are unchanged: results are bit-identical to k_photo_fused.  Measured (see the launch below): it pays for
// a single view, not for a batch - the two stages wait for each other at the row barrier and the first is twice the second.
#ifndef T4D_PH_SPLIT_WAVES
#define T4D_PH_SPLIT_WAVES 4
#endif
template <int kFT>
__global__ __launch_bounds__(2 * kFT) __attribute__((amdgpu_waves_per_eu(T4D_PH_SPLIT_WAVES, T4D_PH_SPLIT_WAVES))) void k_photo_split(const PhP P)
{
    constexpr int kInW = kFT + 2 * kR;
    __shared__ float2 s_in[2][kInW];
    __shared__ float2 s_d01[2][kFT + 2 * kR + 2];
    __shared__ float s_d2[2][kFT + 2 * kR + 2];
    __shared__ float s_red[8];
    const bool first = threadIdx.x < kFT;                // wave-uniform: kFT is a multiple of 64
    const int tid = first ? threadIdx.x : threadIdx.x - kFT;
    const int vc = blockIdx.z, v = vc / 3;
    const int xs = blockIdx.x * P.tw, xe = min(xs + P.tw, P.W);
    const int y0 = blockIdx.y * P.th, y1 = min(y0 + P.th, P.H);
    const size_t HW = (size_t)P.H * P.W;
    const float *im = P.im + (size_t)vc * HW, *gt = P.gt + (size_t)vc * HW;
    const float em = P.cam_m ? expf(P.cam_m[vc]) : 1.f, cc = P.cam_c ? P.cam_c[vc] : 0.f;
    const float N = 3.f * (float)HW, wv = P.weight ? P.weight[v] : 1.f;
    const float g = -0.2f * wv / N;
    const float l1w = 0.8f * wv / N;
    float w[11];
#pragma unroll
    for (int k = 0; k < 11; k++) w[k] = P.win[k];
    for (int b = 0; b < 2; b++) {                        // the first iteration's second stage reads a row nobody wrote
        if (first) {
            s_d01[b][tid] = make_float2(0.f, 0.f); s_d2[b][tid] = 0.f;
            if (tid < 2 * kR + 2) { s_d01[b][kFT + tid] = make_float2(0.f, 0.f); s_d2[b][kFT + tid] = 0.f; }
        }
    }
    const int i_first = y0 - 2 * kR, i_last = y1 + 2 * kR;
    float sum_l1 = 0.f, sum_s = 0.f, sum_gm = 0.f, sum_gc = 0.f;

    if (first) {
        // ---------------- first stage ----------------
        const int gx1 = xs - kR + tid;
        const bool col1 = gx1 >= 0 && gx1 < P.W;
        const bool own1 = tid >= kR && gx1 < xe;
        const int lx0 = xs - 2 * kR + tid, lx1 = lx0 + kFT;
        const bool l0 = lx0 >= 0 && lx0 < P.W, l1 = tid < 2 * kR && lx1 < P.W;
        v2f h01[11], h23[11];
        float h4[11];
#pragma unroll
        for (int k = 0; k < 11; k++) { h01[k] = h23[k] = (v2f){ 0.f, 0.f }; h4[k] = 0.f; }
        float pa0 = 0.f, pb0 = 0.f, pa1 = 0.f, pb1 = 0.f;
        auto fetch = [&](int i) {
            pa0 = pb0 = pa1 = pb1 = 0.f;
            if (i >= 0 && i < P.H) {
                const size_t o = (size_t)i * P.W;
                if (l0) { pa0 = im[o + lx0]; pb0 = gt[o + lx0]; }
                if (l1) { pa1 = im[o + lx1]; pb1 = gt[o + lx1]; }
            }
        };
        fetch(i_first);
        auto row = [&](const int i, auto J_) {
            constexpr int J = decltype(J_)::value;
            const int buf = i & 1;
            {
                const bool in_img = i >= 0 && i < P.H;
                s_in[buf][tid] = (in_img && l0) ? make_float2(em * pa0 + cc, pb0) : make_float2(0.f, 0.f);
                if (tid < 2 * kR) s_in[buf][tid + kFT] = (in_img && l1) ? make_float2(em * pa1 + cc, pb1) : make_float2(0.f, 0.f);
            }
            fetch(i + 1);
            __syncthreads();
            {
                v2f a01 = { 0.f, 0.f }, a23 = { 0.f, 0.f };
                float a4 = 0.f;
#pragma unroll
                for (int k = 0; k < 11; k++) {
                    const v2f ab = *reinterpret_cast<const v2f *>(&s_in[buf][tid + k]);
                    const v2f wab = w[k] * ab;
                    a01 += wab;
                    a23 = __builtin_elementwise_fma(wab, ab, a23);
                    a4 = fmaf(wab.x, ab.y, a4);
                }
                h01[J] = a01; h23[J] = a23; h4[J] = a4;
            }
            v2f m12 = { 0.f, 0.f }, eac = { 0.f, 0.f };
            float eb = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) {
                const int sl = (J + 1 + k) % 11;
                const v2f wk = { w[k], w[k] };
                m12 = __builtin_elementwise_fma(wk, h01[sl], m12);
                eac = __builtin_elementwise_fma(wk, h23[sl], eac);
                eb = fmaf(w[k], h4[sl], eb);
            }
            const float mu1 = m12.x, mu2 = m12.y, ea = eac.x, ec = eac.y;
            const int srow = i - kR;
            float d1 = 0.f, d2 = 0.f, d3 = 0.f;
            if (col1 && srow >= 0 && srow < P.H) {
                const float mu1s = mu1 * mu1, mu2s = mu2 * mu2, mu12 = mu1 * mu2;
                const float s11 = ea - mu1s, s22 = ec - mu2s, s12 = eb - mu12;
                const float A1 = 2.f * mu12 + kC1, A2 = 2.f * s12 + kC2, B1 = mu1s + mu2s + kC1, B2 = s11 + s22 + kC2;
                const float inv = 1.f / (B1 * B2);
                const float S = A1 * A2 * inv;
                const float gi = g * inv;
                d1 = 2.f * gi * (mu2 * (A2 - A1) - S * mu1 * (B2 - B1));
                d2 = -gi * S * B1;
                d3 = 2.f * gi * A1;
                if (own1 && srow >= y0 && srow < y1) sum_s += S;
            }
            s_d01[buf][tid] = make_float2(d1, d2); s_d2[buf][tid] = d3;
        };
        for (int base = i_first; base <= i_last; base += 11) {
            if (base + 0 <= i_last) row(base + 0, IC<0>()); if (base + 1 <= i_last) row(base + 1, IC<1>());
            if (base + 2 <= i_last) row(base + 2, IC<2>()); if (base + 3 <= i_last) row(base + 3, IC<3>());
            if (base + 4 <= i_last) row(base + 4, IC<4>()); if (base + 5 <= i_last) row(base + 5, IC<5>());
            if (base + 6 <= i_last) row(base + 6, IC<6>()); if (base + 7 <= i_last) row(base + 7, IC<7>());
            if (base + 8 <= i_last) row(base + 8, IC<8>()); if (base + 9 <= i_last) row(base + 9, IC<9>());
            if (base + 10 <= i_last) row(base + 10, IC<10>());
        }
    } else {
        // ---------------- second stage ----------------
        const int gx2 = xs + tid;
        const bool col2 = gx2 < xe;
        v2f hd01[11];
        float hd2[11];
#pragma unroll
        for (int k = 0; k < 11; k++) { hd01[k] = (v2f){ 0.f, 0.f }; hd2[k] = 0.f; }
        auto row = [&](const int i, auto J_) {
            constexpr int J = decltype(J_)::value;
            const int buf = i & 1;
            // the pixel of the OUTPUT row of this iteration (eleven rows behind): needed at the very end, requested now
            const int o_row = i - 2 * kR - 1;
            const bool emit = col2 && o_row >= y0 && o_row < y1;
            float o_im = 0.f, o_gt = 0.f;
            if (emit) { o_im = im[(size_t)o_row * P.W + gx2]; o_gt = gt[(size_t)o_row * P.W + gx2]; }
            __syncthreads();
            // the adjoint row written ONE iteration ago by the first stage (made visible by this iteration's barrier)
            const int pb = buf ^ 1;
            v2f q01 = { 0.f, 0.f };
            float q2 = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) {
                const v2f wk = { w[k], w[k] };
                q01 = __builtin_elementwise_fma(wk, *reinterpret_cast<const v2f *>(&s_d01[pb][tid + k]), q01);
                q2 = fmaf(w[k], s_d2[pb][tid + k], q2);
            }
            constexpr int J2 = (J + 10) % 11;
            hd01[J2] = q01; hd2[J2] = q2;
            v2f r01 = { 0.f, 0.f };
            float r2 = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) {
                const int sl = (J2 + 1 + k) % 11;
                const v2f wk = { w[k], w[k] };
                r01 = __builtin_elementwise_fma(wk, hd01[sl], r01);
                r2 = fmaf(w[k], hd2[sl], r2);
            }
            const float r0 = r01.x, r1 = r01.y;
            if (emit) {
                const float x = em * o_im + cc, y = o_gt;
                const float d = x - y;
                sum_l1 += fabsf(d);
                const float gl1 = l1w * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
                const float gg = r0 + 2.f * x * r1 + y * r2 + gl1;
                P.dL_dim[(size_t)vc * HW + (size_t)o_row * P.W + gx2] = em * gg;
                sum_gm += gg * (em * o_im);
                sum_gc += gg;
            }
        };
        for (int base = i_first; base <= i_last; base += 11) {
            if (base + 0 <= i_last) row(base + 0, IC<0>()); if (base + 1 <= i_last) row(base + 1, IC<1>());
            if (base + 2 <= i_last) row(base + 2, IC<2>()); if (base + 3 <= i_last) row(base + 3, IC<3>());
            if (base + 4 <= i_last) row(base + 4, IC<4>()); if (base + 5 <= i_last) row(base + 5, IC<5>());
            if (base + 6 <= i_last) row(base + 6, IC<6>()); if (base + 7 <= i_last) row(base + 7, IC<7>());
            if (base + 8 <= i_last) row(base + 8, IC<8>()); if (base + 9 <= i_last) row(base + 9, IC<9>());
            if (base + 10 <= i_last) row(base + 10, IC<10>());
        }
    }
    const float tl1 = block_sum(sum_l1, s_red), tss = block_sum(sum_s, s_red);
    const float tgm = block_sum(sum_gm, s_red), tgc = block_sum(sum_gc, s_red);
    if (threadIdx.x == 0) {
        const size_t t = ((size_t)vc * P.ty + blockIdx.y) * P.tx + blockIdx.x;
        P.part_loss[2 * t] = tl1; P.part_loss[2 * t + 1] = tss;
        P.part_cam[2 * t] = tgm; P.part_cam[2 * t + 1] = tgc;
    }
}

// one workgroup per view: fixed-order sums of the per-tile partials
__global__ __launch_bounds__(kBlock) void k_photo_final(const PhP P)
{
    __shared__ float s_red[4];
    const int v = blockIdx.x, tid = threadIdx.x;
    const int tiles = P.tx * P.ty;
    float l1 = 0.f, ss = 0.f;
    for (int i = tid; i < 3 * tiles; i += kBlock) {
        const size_t t = (size_t)v * 3 * tiles + i;
        l1 += P.part_loss[2 * t]; ss += P.part_loss[2 * t + 1];
    }
    const float tl1 = block_sum(l1, s_red), tss = block_sum(ss, s_red);
    const float N = 3.f * (float)P.H * (float)P.W;
    if (tid == 0) P.loss[v] = 0.8f * (tl1 / N) + 0.2f * (1.f - tss / N);
    if (P.dL_dm && P.dL_dc) {
        for (int ch = 0; ch < 3; ch++) {
            float gm = 0.f, gc = 0.f;
            for (int i = tid; i < tiles; i += kBlock) {
                const size_t t = ((size_t)v * 3 + ch) * tiles + i;
                gm += P.part_cam[2 * t]; gc += P.part_cam[2 * t + 1];
            }
            const float tgm = block_sum(gm, s_red), tgc = block_sum(gc, s_red);
            if (tid == 0) { P.dL_dm[v * 3 + ch] = tgm; P.dL_dc[v * 3 + ch] = tgc; }
        }
    }
}

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// ---------------------------------------------------------------------------------------------------------
// Masked L1 of the dense (texture) pass, reference train.py:394-405 (get_loss_dense, use_mask=True):
//     masked_index = filtered_mask == 1;   loss = sum_{masked} |im - gt| / masked_index.sum()
// (no camera affine, no SSIM; filtered_mask = helpers.get_mask(...), a float [3,H,W] image of zeros and ones).
// Two launches per batch of views: partial sums + counts, then a fixed-order total per view and the gradient
// dL/dim = weight * sign(im - gt) / count on the masked elements, 0 elsewhere.
// ---------------------------------------------------------------------------------------------------------
constexpr int kMlBlocks = 128;            // partial sums per view

__global__ __launch_bounds__(kBlock) void k_masked_l1_partial(const float *im, const float *gt, const float *mask, size_t n,
                                                              float *part_sum, uint32_t *part_cnt)
{
    __shared__ float s_red[4];
    __shared__ uint32_t s_cnt[4];
    const int v = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
    const size_t base = (size_t)v * n;
    float acc = 0.f;
    uint32_t cnt = 0;
    for (size_t i = (size_t)blk * kBlock + tid; i < n; i += (size_t)kMlBlocks * kBlock) {
        if (mask[base + i] == 1.0f) { acc += fabsf(im[base + i] - gt[base + i]); cnt++; }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { acc += __shfl_xor(acc, d, 64); cnt += (uint32_t)__shfl_xor((int)cnt, d, 64); }
    if ((tid & 63) == 0) { s_red[tid >> 6] = acc; s_cnt[tid >> 6] = cnt; }
    __syncthreads();
    if (tid == 0) {
        part_sum[(size_t)v * kMlBlocks + blk] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
        part_cnt[(size_t)v * kMlBlocks + blk] = (s_cnt[0] + s_cnt[1]) + (s_cnt[2] + s_cnt[3]);
    }
}

__global__ __launch_bounds__(kBlock) void k_masked_l1_grad(const float *im, const float *gt, const float *mask, size_t n,
                                                           const float *part_sum, const uint32_t *part_cnt, const float *weight,
                                                           float *loss, float *dL_dim)
{
    __shared__ float s_tot;
    __shared__ float s_count;
    const int v = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
    if (tid < 64) {                                                   // fixed-order total of the view's partials
        float a = part_sum[(size_t)v * kMlBlocks + tid] + part_sum[(size_t)v * kMlBlocks + 64 + tid];
        unsigned long long c = (unsigned long long)part_cnt[(size_t)v * kMlBlocks + tid] + part_cnt[(size_t)v * kMlBlocks + 64 + tid];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            a += __shfl_xor(a, d, 64);
            c += ((unsigned long long)(uint32_t)__shfl_xor((int)(c >> 32), d, 64) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)c, d, 64);
        }
        if (tid == 0) { s_tot = a; s_count = (float)c; }
    }
    __syncthreads();
    const float count = s_count;
    if (blk == 0 && tid == 0) loss[v] = s_tot / count;                // 0 / 0 = NaN, like the reference's empty mask
    const float g = (weight ? weight[v] : 1.f) / count;
    const size_t base = (size_t)v * n;
    for (size_t i = (size_t)blk * kBlock + tid; i < n; i += (size_t)gridDim.x * kBlock) {
        float o = 0.f;
        if (mask[base + i] == 1.0f) {
            const float d = im[base + i] - gt[base + i];
            o = d > 0.f ? g : (d < 0.f ? -g : 0.f);
        }
        dL_dim[base + i] = o;
    }
}

}  // namespace

T4D_EXPORT size_t t4d_masked_l1_scratch_bytes(int32_t n_views)
{
    return n_views < 1 ? 0 : 2 * align_up((size_t)n_views * kMlBlocks * 4);
}

T4D_EXPORT int t4d_masked_l1_loss(int32_t n_views, int32_t H, int32_t W, const float *im, const float *gt, const float *mask,
                                  const float *view_weight, float *loss, float *dL_dim, void *scratch, size_t scratch_bytes,
                                  void *hip_stream)
{
    static_assert(kMlBlocks == 128, "k_masked_l1_grad sums two partials per lane of one wave");
    if (n_views < 1 || H < 1 || W < 1 || !im || !gt || !mask || !loss || !dL_dim || !scratch)
        return t4d_internal_fail(T4D_ERR_ARG, "t4d_masked_l1_loss: bad arguments%s", "");
    if (n_views > 65535) return t4d_internal_fail(T4D_ERR_ARG, "t4d_masked_l1_loss: too many views%s", "");
    if (scratch_bytes < t4d_masked_l1_scratch_bytes(n_views))
        return t4d_internal_fail(T4D_ERR_STATE_SIZE, "t4d_masked_l1_loss: scratch too small%s", "");
    const size_t n = (size_t)3 * H * W;
    float *part_sum = (float *)scratch;
    uint32_t *part_cnt = (uint32_t *)((char *)scratch + align_up((size_t)n_views * kMlBlocks * 4));
    hipStream_t stream = (hipStream_t)hip_stream;
    hipLaunchKernelGGL(k_masked_l1_partial, dim3(kMlBlocks, n_views), dim3(kBlock), 0, stream, im, gt, mask, n, part_sum, part_cnt);
    const unsigned gblocks = (unsigned)((n + (size_t)kBlock * 8 - 1) / ((size_t)kBlock * 8));
    hipLaunchKernelGGL(k_masked_l1_grad, dim3(gblocks < 1 ? 1 : gblocks, n_views), dim3(kBlock), 0, stream, im, gt, mask, n,
                       (const float *)part_sum, (const uint32_t *)part_cnt, view_weight, loss, dL_dim);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return t4d_internal_fail(T4D_ERR_HIP, "t4d_masked_l1_loss launch: %s", hipGetErrorString(e));
    return T4D_OK;
}

// strips and row segments of the fused kernel for an H x W image: strips as even as possible; segments of 128 rows from 16 M
// values per batch on, 64 from 2 M, 16 below (one view of Topo4D's 512 x 375 images: more, shorter workgroups - every segment
// pays 21 warm-up rows, but a lone view is latency-bound)
static void photo_tiling(int32_t n_views, int32_t H, int32_t W, int *tx, int *ty, int *tw, int *th, int *threads)
{
    int best = 0;
    long long best_cost = 0;
    for (int ft = 64; ft <= 256; ft += 64) {
        const int strips = (W + (ft - 2 * kR) - 1) / (ft - 2 * kR);
        const long long cost = (long long)strips * ft;
        if (best == 0 || cost < best_cost) { best = ft; best_cost = cost; }
    }
    if (const char *e = getenv("T4D_PH_THREADS")) { const int ft = atoi(e); if (ft == 64 || ft == 128 || ft == 192 || ft == 256) best = ft; }
    *threads = best;
    *tx = (W + (best - 2 * kR) - 1) / (best - 2 * kR);
    *tw = (W + *tx - 1) / *tx;
    const long long work = (long long)n_views * 3 * H * W;
    *th = work >= (1ll << 24) ? 128 : (work >= (1ll << 21) ? 64 : 16);
    if (const char *e = getenv("T4D_PH_ROWS")) *th = atoi(e) > 0 ? atoi(e) : *th;      // experiments
    if (*th > H) *th = H;
    *ty = (H + *th - 1) / *th;
}

T4D_EXPORT size_t t4d_photometric_scratch_bytes(int32_t n_views, int32_t H, int32_t W)
{
    if (n_views < 1 || H < 1 || W < 1) return 0;
    int tx, ty, tw, th, ft;
    photo_tiling(n_views, H, W, &tx, &ty, &tw, &th, &ft);
    const size_t tiles = (size_t)tx * ty * n_views * 3;
    return 2 * align_up(tiles * 8);
}

T4D_EXPORT int t4d_photometric_loss(int32_t n_views, int32_t H, int32_t W, const float *im, const float *gt, const float *cam_m,
                                    const float *cam_c, const float *view_weight, float *loss, float *dL_dim, float *dL_dcam_m,
                                    float *dL_dcam_c, void *scratch, size_t scratch_bytes, void *hip_stream)
{
    if (n_views < 1 || H < 1 || W < 1 || !im || !gt || !loss || !dL_dim || !scratch)
        return t4d_internal_fail(T4D_ERR_ARG, "t4d_photometric_loss: bad arguments%s", "");
    if ((cam_m == nullptr) != (cam_c == nullptr) || (dL_dcam_m == nullptr) != (dL_dcam_c == nullptr))
        return t4d_internal_fail(T4D_ERR_ARG, "t4d_photometric_loss: cam_m/cam_c (and their gradients) come in pairs%s", "");
    if (scratch_bytes < t4d_photometric_scratch_bytes(n_views, H, W))
        return t4d_internal_fail(T4D_ERR_STATE_SIZE, "t4d_photometric_loss: scratch too small%s", "");
    if ((size_t)n_views * 3 > 65535) return t4d_internal_fail(T4D_ERR_ARG, "t4d_photometric_loss: too many views%s", "");
    PhP P;
    memset(&P, 0, sizeof(P));
    P.V = n_views; P.H = H; P.W = W;
    int ft = 0;
    photo_tiling(n_views, H, W, &P.tx, &P.ty, &P.tw, &P.th, &ft);
    if (P.ty > 65535) return t4d_internal_fail(T4D_ERR_ARG, "t4d_photometric_loss: image too tall%s", "");
    P.im = im; P.gt = gt; P.cam_m = cam_m; P.cam_c = cam_c; P.weight = view_weight;
    P.loss = loss; P.dL_dim = dL_dim; P.dL_dm = dL_dcam_m; P.dL_dc = dL_dcam_c;
    const size_t tiles = (size_t)P.tx * P.ty * n_views * 3;
    char *sc = (char *)scratch;
    P.part_loss = (float *)sc;
    P.part_cam = (float *)(sc + align_up(tiles * 8));
    // the reference's window: exp(-(i-5)^2 / (2*1.5^2)) for i = 0..10, as float32, normalised in float32 (external.py:73-76)
    float g[11], sum = 0.f;
    for (int i = 0; i < 11; i++) { g[i] = (float)exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5)); sum += g[i]; }
    for (int i = 0; i < 11; i++) P.win[i] = g[i] / sum;
    hipStream_t stream = (hipStream_t)hip_stream;
    const dim3 grid(P.tx, P.ty, n_views * 3);
    // Two wave roles per strip (k_photo_split) for ONE view's worth of pixels - a launch that cannot fill the chip, where the
    // second set of waves shortens every workgroup's row: 36.5 -> 29.7 us for a 512 x 375 view.  A batch of views keeps one
    // thread per column for both stages: there the roles only add waves that wait for each other at the row barrier (the first
    // stage is twice the second: 24 x 512^2 235 -> 258-330 us, 24 x 2048^2 -2 %).  T4D_PH_SPLIT=0/1 forces one or the other.
    const char *split_env = getenv("T4D_PH_SPLIT");
    const bool split = split_env ? atoi(split_env) != 0 : (long long)n_views * 3 * H * W < (1ll << 21);
    if (split) {
        if (ft == 64) hipLaunchKernelGGL(k_photo_split<64>, grid, dim3(128), 0, stream, P);
        else if (ft == 128) hipLaunchKernelGGL(k_photo_split<128>, grid, dim3(256), 0, stream, P);
        else if (ft == 192) hipLaunchKernelGGL(k_photo_split<192>, grid, dim3(384), 0, stream, P);
        else hipLaunchKernelGGL(k_photo_split<256>, grid, dim3(512), 0, stream, P);
    } else if (ft == 64) hipLaunchKernelGGL(k_photo_fused<64>, grid, dim3(64), 0, stream, P);
    else if (ft == 128) hipLaunchKernelGGL(k_photo_fused<128>, grid, dim3(128), 0, stream, P);
    else if (ft == 192) hipLaunchKernelGGL(k_photo_fused<192>, grid, dim3(192), 0, stream, P);
    else hipLaunchKernelGGL(k_photo_fused<256>, grid, dim3(256), 0, stream, P);
    hipLaunchKernelGGL(k_photo_final, dim3(n_views), dim3(kBlock), 0, stream, P);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return t4d_internal_fail(T4D_ERR_HIP, "t4d_photometric_loss launch: %s", hipGetErrorString(e));
    return T4D_OK;
}
