// t4d_photometric.hip — fused photometric loss (forward + gradient) for Topo4D's render loop on MI355X
// (SURVEY.md §8 row a13 / §8f rank 1).
//
// Restates reference train.py:310,315 with helpers.py:115-116 (`l1_loss_v1`) and external.py:73-116 (`calc_ssim`):
//     im' = exp(cam_m[c]) * im + cam_c[c]
//     loss_v = 0.8 * mean|im' - gt| + 0.2 * (1 - mean SSIM_11x11(im', gt))          (mean over the 3*H*W values of view v)
// SSIM: depthwise 11x11 Gaussian window (sigma 1.5), zero padding, c1 = 0.01^2, c2 = 0.03^2.
// The reference spends 5 conv2d launches forward and their autograd backward per iteration; here one launch set per
// batch of V views produces the per-view losses AND dL/d(im) (the rasterizer's backward input) AND dL/d(cam_m, cam_c):
//   k_photo_stream  one workgroup per (view, channel, strip of columns, segment of rows) streams down its strip: separable
//                   11-tap filter of (x, y, x^2 + y^2, x y), SSIM + L1 -> partial loss sums and the three adjoint values
//                   D1 = g dS/dmu1, D2 = g dS/dE[x^2], D3 = g dS/dE[xy] of the row five behind; separable filter of those
//                   (the window is symmetric: adjoint = same filter) -> dL/dx' = G*D1 + 2x'(G*D2) + y(G*D3) + L1 term of
//                   the row eleven behind; affine backward; partial sums for cam_m / cam_c.  No intermediate in memory.
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

constexpr int kR = 5;              // window radius (11 taps)
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
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));      // packed-math pair: one v_pk_fma_f32 does two of the filter's multiply-adds

// The per-pixel arithmetic both loss kernels share (one source, one rounding): SSIM of the four filtered maps and the three
// adjoint values, and dL/dx' of an output pixel.
//   S = A1 A2 / (B1 B2),  s11 + s22 = E[x^2 + y^2] - mu1^2 - mu2^2,  s12 = E[xy] - mu1 mu2;  gsc = dL/dS inside the image, else 0
__device__ __forceinline__ void ssim_adjoint(const float mu1, const float mu2, const float es, const float ep, const float gsc,
                                             float &S, float &d1, float &d2, float &d3)
{
    const float mu12 = mu1 * mu2;
    const float B1 = fmaf(mu1, mu1, fmaf(mu2, mu2, kC1));
    const float B2 = (es - B1) + (kC1 + kC2);
    const float A1 = fmaf(2.f, mu12, kC1), A2 = fmaf(2.f, ep - mu12, kC2);
    const float den = B1 * B2;
    float inv = __builtin_amdgcn_rcpf(den);              // the one reciprocal of the pixel: 1/B1 = B2 inv, 1/B2 = B1 inv
    inv = fmaf(fmaf(-den, inv, 1.f), inv, inv);
    S = A1 * A2 * inv;
    const float gi = gsc * inv;
    const float gi2 = gi + gi;
    d1 = gi2 * (mu2 * (A2 - A1) - S * mu1 * (B2 - B1));
    d2 = -gi * S * B1;
    d3 = gi2 * A1;
}
// x' = em * im + cc, y = gt, (r0, r1, r2) = G*D1, G*D2, G*D3 at the pixel: returns dL/dx', leaves |x' - y| in `ad`
__device__ __forceinline__ float pixel_gradient(const float x, const float y, const float r0, const float r1, const float r2,
                                                const float l1w, float &ad)
{
    const float d = x - y;
    ad = fabsf(d);
    const float gl1 = d == 0.f ? 0.f : copysignf(l1w, d);                        // 0.8 / N * sign(x' - y)
    return fmaf(x + x, r1, r0) + fmaf(y, r2, gl1);
}

// The strip kernel.  A workgroup of kFT threads owns a vertical STRIP of one channel of one view - kFT input columns, hence
// kFT - 10 first-stage columns and kFT - 20 output columns - and a segment of rows, and streams down it one image row per
// iteration; every thread owns ONE column of each stage:
//   * the thread that loads a pixel forms (x', y, s = x'^2 + y^2, p = x' y) ONCE and writes them to LDS as one 16-byte entry;
//     a thread's 11-tap HORIZONTAL sums of the four maps are eleven ds_read_b128 and twenty-two v_pk_fma_f32.  (SSIM needs
//     E[x^2] and E[y^2] only as their sum: B2 = E[x^2] + E[y^2] - mu1^2 - mu2^2 + c2, so four maps suffice where round 3
//     filtered five and squared every pixel once per tap.)  The sums go into a window of the last eleven rows that lives in
//     REGISTERS; the VERTICAL sums over it are the filtered maps at the row five behind - SSIM, loss terms and the three
//     adjoint values of that pixel, one reciprocal (+ a Newton step) per pixel;
//   * the adjoint row goes to LDS; each thread takes its horizontal sums, pushes them into a second register window, and the
//     vertical sums over that give G*D1, G*D2, G*D3 at the row five further behind: dL/dx' and the affine backward.  The
//     output pixel is fetched again (an L2 hit) rather than carried through eleven iterations.
// Rows are unrolled eleven at a time so that the window slots are compile-time register names (no shifting).  One barrier per
// row; no other branch in the row body than two wave-uniform row tests: loads use clamped addresses through buffer
// descriptors (column offset in a loop-invariant register, row offset in a scalar: no vector address arithmetic), validity
// is a select, columns that belong to a neighbouring strip are computed and dropped from the sums at the very end.
// 123 registers: four waves per SIMD.  Round 3's kernel (five maps, (x', y) pairs in LDS, IEEE division, 25 branches per
// row: ~250 vector instructions per thread and row) -> ~160: 24 x 512^2 224 -> 180 us, 24 x 2048^2 2.95 -> 2.29 ms.  What it is
// bound by (ablation builds, tools/experiments/README.md): without its LDS reads and barriers it still takes 162 us - the
// vector ALUs of the busiest CUs (864 workgroups on 256 CUs: four on some, three on others).
template <int kFT>
__global__ __launch_bounds__(kFT) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_photo_stream(const PhP P)
{
    constexpr int kRow = kFT + 16;                        // entries per LDS row: kFT + the ten a last thread reads beyond
    __shared__ v4f s_in[2][kRow];                         // (x', y, x'^2 + y^2, x' y) of the current input row
    __shared__ v4f s_d[2][kRow];                          // (D1, D2, D3, -) of the current adjoint row
    __shared__ float s_red[4];
    const int tid = threadIdx.x;
    const int vc = blockIdx.z, v = vc / 3;
    const int xs = blockIdx.x * P.tw, xe = min(xs + P.tw, P.W);              // output columns [xs, xe)
    const int y0 = blockIdx.y * P.th, y1 = min(y0 + P.th, P.H);              // output rows [y0, y1)
    const size_t HW = (size_t)P.H * P.W;
    // one buffer descriptor per plane: a load is `buffer_load_dword v, v_col4, s[rsrc], s_row_offset offen`
    const unsigned plane_bytes = (unsigned)(HW * 4);
    const __amdgpu_buffer_rsrc_t r_im = __builtin_amdgcn_make_buffer_rsrc((void *)(P.im + (size_t)vc * HW), 0, plane_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_gt = __builtin_amdgcn_make_buffer_rsrc((void *)(P.gt + (size_t)vc * HW), 0, plane_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc((void *)(P.dL_dim + (size_t)vc * HW), 0, plane_bytes, 0x00020000);
    auto ldf = [](__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
    };
    const float em = P.cam_m ? expf(P.cam_m[vc]) : 1.f, cc = P.cam_c ? P.cam_c[vc] : 0.f;
    const float N = 3.f * (float)HW, wv = P.weight ? P.weight[v] : 1.f;
    const float g = -0.2f * wv / N;                      // dL/dS
    const float l1w = 0.8f * wv / N;
    float w[11];
#pragma unroll
    for (int k = 0; k < 11; k++) w[k] = P.win[k];

    // this thread's input column, first-stage column and output column
    const int gin = xs - 2 * kR + tid, g1 = xs - kR + tid, g2 = xs + tid;
    const bool in_col = gin >= 0 && gin < P.W;
    const bool col1 = g1 >= 0 && g1 < P.W;               // the adjoint map exists inside the image only
    const bool own1 = g1 >= xs && g1 < xe;               // ... and this strip counts its SSIM term
    const bool col2 = g2 < xe;                           // an output column of this strip
    const unsigned off_in = 4u * (unsigned)min(max(gin, 0), P.W - 1), off_out = 4u * (unsigned)min(g2, P.W - 1);
    // zero both buffers once (the entries beyond the strip's columns stay zero)
    for (int e = tid; e < kRow; e += kFT) {
        s_in[0][e] = s_in[1][e] = (v4f){ 0.f, 0.f, 0.f, 0.f };
        s_d[0][e] = s_d[1][e] = (v4f){ 0.f, 0.f, 0.f, 0.f };
    }
    // the two register windows (slots are compile-time indices): (x, y), (s, p) and (D1, D2) | D3
    v2f h01[11], h23[11], hd01[11];
    float hd2[11];
#pragma unroll
    for (int k = 0; k < 11; k++) {
        h01[k] = h23[k] = hd01[k] = (v2f){ 0.f, 0.f };
        hd2[k] = 0.f;
    }
    // partial sums: rows are gated by wave-uniform tests inside the loop, this thread's columns once at the end
    float sum_l1 = 0.f, sum_s = 0.f, sum_gm = 0.f, sum_gc = 0.f;
    const int i_first = y0 - 2 * kR, i_last = y1 + 2 * kR;      // input rows i_first .. i_last (the last ones only drain)

    float pa, pb;                                        // the next row's pixel, in flight
    auto fetch = [&](int i) {
        const unsigned ro = (unsigned)min(max(i, 0), P.H - 1) * (unsigned)P.W * 4u;
        pa = ldf(r_im, off_in, ro); pb = ldf(r_gt, off_in, ro);
    };
    fetch(i_first);
    __syncthreads();

    auto row = [&](const int i, auto J_) {
        constexpr int J = decltype(J_)::value;           // slot of this row in the first window
        const int buf = i & 1;
        // ---- this row's inputs to LDS, zero outside the image (external.py:86 padding=5); next row's loads in flight
        {
            const bool ok = (unsigned)i < (unsigned)P.H && in_col;
            const float x = ok ? fmaf(em, pa, cc) : 0.f, y = ok ? pb : 0.f;
            s_in[buf][tid] = (v4f){ x, y, fmaf(x, x, y * y), x * y };
        }
        fetch(i + 1);
        // the pixel of this iteration's OUTPUT row (eleven rows behind): needed at the very end, requested now
        const int o_row = i - 2 * kR - 1;
        const bool o_ok = o_row >= y0 && o_row < y1;
        const unsigned oro = (unsigned)min(max(o_row, 0), P.H - 1) * (unsigned)P.W * 4u;
        const float o_im = ldf(r_im, off_out, oro), o_gt = ldf(r_gt, off_out, oro);
        __syncthreads();
        // ---- first stage, horizontal
        {
            v2f a01 = { 0.f, 0.f }, a23 = { 0.f, 0.f };
#pragma unroll
            for (int k = 0; k < 11; k++) {
                const v4f q = s_in[buf][tid + k];
                const v2f wk = { w[k], w[k] };
                a01 = __builtin_elementwise_fma(wk, (v2f){ q.x, q.y }, a01);
                a23 = __builtin_elementwise_fma(wk, (v2f){ q.z, q.w }, a23);
            }
            h01[J] = a01; h23[J] = a23;
        }
        // ---- first stage, vertical: rows i-10 .. i are in slots J+1 .. J+11 (mod 11) -> the filtered maps at row i - 5
        {
            v2f m12 = { 0.f, 0.f }, esp = { 0.f, 0.f };
#pragma unroll
            for (int k = 0; k < 11; k++) {
                const int sl = (J + 1 + k) % 11;
                const v2f wk = { w[k], w[k] };
                m12 = __builtin_elementwise_fma(wk, h01[sl], m12);
                esp = __builtin_elementwise_fma(wk, h23[sl], esp);
            }
            const int srow = i - kR;
            float S, d1, d2, d3;
            ssim_adjoint(m12.x, m12.y, esp.x, esp.y, ((unsigned)srow < (unsigned)P.H && col1) ? g : 0.f, S, d1, d2, d3);
            s_d[buf][tid] = (v4f){ d1, d2, d3, 0.f };
            if (srow >= y0 && srow < y1) sum_s += S;     // wave-uniform
        }
        // ---- second stage on the adjoint row written ONE iteration ago (made visible by this iteration's barrier)
        {
            const int pb_ = buf ^ 1;
            v2f q01 = { 0.f, 0.f };
            float q2 = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) {
                // (the empty asm keeps the compiler from narrowing the read to the three values used: ds_read_b96 runs at
                // 96 B/clk, ds_read_b128 at 256)
                v4f q = s_d[pb_][tid + k];
                asm("" : "+v"(q));
                const v2f wk = { w[k], w[k] };
                q01 = __builtin_elementwise_fma(wk, (v2f){ q.x, q.y }, q01);
                q2 = fmaf(w[k], q.z, q2);
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
            if (o_ok) {                                  // wave-uniform: an output row of this segment
                const float xi = em * o_im;
                float ad;
                const float gg = pixel_gradient(fmaf(em, o_im, cc), o_gt, r01.x, r01.y, r2, l1w, ad);      // dL/dx'
                if (col2) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, em * gg), r_out, off_out, oro, 0);
                sum_l1 += ad;
                sum_gm = fmaf(gg, xi, sum_gm);           // d x'/d cam_m = exp(cam_m) * im
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
    const float tl1 = block_sum(col2 ? sum_l1 : 0.f, s_red), tss = block_sum(own1 ? sum_s : 0.f, s_red);
    const float tgm = block_sum(col2 ? sum_gm : 0.f, s_red), tgc = block_sum(col2 ? sum_gc : 0.f, s_red);
    if (tid == 0) {
        const size_t t = ((size_t)vc * P.ty + blockIdx.y) * P.tx + blockIdx.x;
        P.part_loss[2 * t] = tl1; P.part_loss[2 * t + 1] = tss;
        P.part_cam[2 * t] = tgm; P.part_cam[2 * t + 1] = tgc;
    }
}

// The TILE kernel for small launches (one to three views of Topo4D's 512 x 375 images: train.py:661-673 computes the loss of ONE
// view per iteration).  A strip workgroup walks 20 warm-up rows plus its segment one row after the other - 36 dependent rows of
// ~0.85 us for a lone wave: 31 us for a view whose arithmetic is worth 6.  Here a workgroup of NT threads owns a tile of
// TH x 44 output pixels and runs the four filter passes as four PHASES over the whole tile, every thread taking its share of a
// phase's pixels, with the intermediate maps in LDS instead of register windows:
//   load (TH+20) x 64 inputs -> P1 horizontal 11 taps of (x, y, s, p) -> P2 vertical taps, SSIM, adjoint values (TH+10) x 54
//   -> P3 horizontal taps of (D1, D2, D3) -> P4 vertical taps, dL/dx', sums.
// Four barriers instead of TH + 21, a dependent chain of ~2,000 instructions per thread instead of ~5,800.  The arithmetic and
// its order per pixel are the strip kernel's (ssim_adjoint, pixel_gradient, taps in ascending order): dL/dim is bit-identical
// whichever kernel a launch takes (tests/test_gpu_photometric.py).  LDS: inputs as (x', y) pairs - s and p are formed per tap
// here - overlaid by the adjoint values, the first stage's maps overlaid by the second stage's.  Two shapes: 64 x 44 pixels on
// 1,024 threads (118 KB of LDS: one workgroup per CU, sixteen waves; one 512 x 375 view = 216 tiles) when the launch has at most
// one tile per CU, else 32 x 44 on 512 threads (71 KB: two per CU) up to two tiles per CU; bigger launches take the strips.
template <int TH, int NT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_photo_tile(const PhP P)
{
    constexpr int TW = 44, IW = TW + 4 * kR, MW = TW + 2 * kR, IH = TH + 4 * kR, DH = TH + 2 * kR;
    constexpr int kBytesA = (IH * IW * 8 > DH * MW * 12 ? IH * IW * 8 : DH * MW * 12);
    __shared__ __attribute__((aligned(16))) unsigned char s_a[kBytesA];     // inputs (x', y); then D1 D2 | D3
    __shared__ v4f s_h[IH * MW];                                            // first-stage maps; then G_h*(D1, D2) | G_h*D3
    __shared__ float s_red[NT / 64];
    v2f *s_in = reinterpret_cast<v2f *>(s_a);
    v2f *s_d01 = reinterpret_cast<v2f *>(s_a);
    float *s_d2 = reinterpret_cast<float *>(s_a + DH * MW * 8);
    v2f *s_q01 = reinterpret_cast<v2f *>(s_h);
    float *s_q2 = reinterpret_cast<float *>(s_h) + 2 * DH * TW;
    const int tid = threadIdx.x;
    const int vc = blockIdx.z, v = vc / 3;
    const int x0 = blockIdx.x * TW, x1 = min(x0 + TW, P.W), y0 = blockIdx.y * TH, y1 = min(y0 + TH, P.H);
    const size_t HW = (size_t)P.H * P.W;
    const float *im = P.im + (size_t)vc * HW, *gt = P.gt + (size_t)vc * HW;
    float *dout = P.dL_dim + (size_t)vc * HW;
    const float em = P.cam_m ? expf(P.cam_m[vc]) : 1.f, cc = P.cam_c ? P.cam_c[vc] : 0.f;
    const float N = 3.f * (float)HW, wv = P.weight ? P.weight[v] : 1.f;
    const float g = -0.2f * wv / N, l1w = 0.8f * wv / N;
    float w[11];
#pragma unroll
    for (int k = 0; k < 11; k++) w[k] = P.win[k];
    // ---- load: (x', y) of the tile and its 10-pixel frame, zero outside the image (external.py:86 padding=5).  All of a thread's
    // loads are issued before the first is used (clamped addresses, validity as a select): one round trip, not thirteen
    constexpr int kIn = (IH * IW + NT - 1) / NT;
    {
        float a[kIn], b[kIn];
#pragma unroll
        for (int j = 0; j < kIn; j++) {
            const int i = tid + NT * j, r = i / IW, c = i - r * IW;
            const size_t o = (size_t)min(max(y0 - 2 * kR + r, 0), P.H - 1) * P.W + min(max(x0 - 2 * kR + c, 0), P.W - 1);
            a[j] = im[o]; b[j] = gt[o];
        }
#pragma unroll
        for (int j = 0; j < kIn; j++) {
            const int i = tid + NT * j, r = i / IW, c = i - r * IW, gy = y0 - 2 * kR + r, gx = x0 - 2 * kR + c;
            const bool ok = (unsigned)gy < (unsigned)P.H && (unsigned)gx < (unsigned)P.W;
            if (i < IH * IW) s_in[i] = (v2f){ ok ? fmaf(em, a[j], cc) : 0.f, ok ? b[j] : 0.f };
        }
    }
    // the output pixels of this thread (phase 4), requested now
    constexpr int kOut = (TH * TW + NT - 1) / NT;
    float o_im[kOut], o_gt[kOut];
#pragma unroll
    for (int j = 0; j < kOut; j++) {
        const int i = tid + NT * j, r = i / TW, c = i - r * TW;
        const int gy = min(y0 + r, P.H - 1), gx = min(x0 + c, P.W - 1);
        o_im[j] = im[(size_t)gy * P.W + gx]; o_gt[j] = gt[(size_t)gy * P.W + gx];
    }
    __syncthreads();
    // ---- P1: horizontal taps of (x, y, s = x^2 + y^2, p = x y) at IH x MW positions
    for (int i = tid; i < IH * MW; i += NT) {
        const int r = i / MW, c = i - r * MW;
        const v2f *src = s_in + r * IW + c;
        v2f a01 = { 0.f, 0.f }, a23 = { 0.f, 0.f };
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const v2f q = src[k];
            const v2f wk = { w[k], w[k] };
            a01 = __builtin_elementwise_fma(wk, q, a01);
            a23 = __builtin_elementwise_fma(wk, (v2f){ fmaf(q.x, q.x, q.y * q.y), q.x * q.y }, a23);
        }
        s_h[i] = (v4f){ a01.x, a01.y, a23.x, a23.y };
    }
    __syncthreads();
    // ---- P2: vertical taps -> filtered maps at DH x MW positions -> SSIM, adjoint values (over the input pairs: all read)
    float sum_s = 0.f;
    for (int i = tid; i < DH * MW; i += NT) {
        const int r = i / MW, c = i - r * MW;
        v2f m12 = { 0.f, 0.f }, esp = { 0.f, 0.f };
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const v4f q = s_h[i + k * MW];
            const v2f wk = { w[k], w[k] };
            m12 = __builtin_elementwise_fma(wk, (v2f){ q.x, q.y }, m12);
            esp = __builtin_elementwise_fma(wk, (v2f){ q.z, q.w }, esp);
        }
        const int gy = y0 - kR + r, gx = x0 - kR + c;
        const bool inside = (unsigned)gy < (unsigned)P.H && (unsigned)gx < (unsigned)P.W;
        float S, d1, d2, d3;
        ssim_adjoint(m12.x, m12.y, esp.x, esp.y, inside ? g : 0.f, S, d1, d2, d3);
        s_d01[i] = (v2f){ d1, d2 }; s_d2[i] = d3;
        sum_s += (gy >= y0 && gy < y1 && gx >= x0 && gx < x1) ? S : 0.f;
    }
    __syncthreads();
    // ---- P3: horizontal taps of (D1, D2, D3) at DH x TW positions (over the first-stage maps: all read)
    for (int i = tid; i < DH * TW; i += NT) {
        const int r = i / TW, c = i - r * TW;
        const int src = r * MW + c;
        v2f q01 = { 0.f, 0.f };
        float q2 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const v2f wk = { w[k], w[k] };
            q01 = __builtin_elementwise_fma(wk, s_d01[src + k], q01);
            q2 = fmaf(w[k], s_d2[src + k], q2);
        }
        s_q01[i] = q01; s_q2[i] = q2;
    }
    __syncthreads();
    // ---- P4: vertical taps -> dL/dx', affine backward, sums
    float sum_l1 = 0.f, sum_gm = 0.f, sum_gc = 0.f;
#pragma unroll
    for (int j = 0; j < kOut; j++) {
        const int i = tid + NT * j, r = i / TW, c = i - r * TW;
        if (i < TH * TW) {
            v2f r01 = { 0.f, 0.f };
            float r2 = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) {
                const v2f wk = { w[k], w[k] };
                r01 = __builtin_elementwise_fma(wk, s_q01[i + k * TW], r01);
                r2 = fmaf(w[k], s_q2[i + k * TW], r2);
            }
            const int gy = y0 + r, gx = x0 + c;
            if (gy < y1 && gx < x1) {
                const float xi = em * o_im[j];
                float ad;
                const float gg = pixel_gradient(fmaf(em, o_im[j], cc), o_gt[j], r01.x, r01.y, r2, l1w, ad);
                dout[(size_t)gy * P.W + gx] = em * gg;
                sum_l1 += ad;
                sum_gm = fmaf(gg, xi, sum_gm);
                sum_gc += gg;
            }
        }
    }
    const float tl1 = block_sum(sum_l1, s_red), tss = block_sum(sum_s, s_red);
    const float tgm = block_sum(sum_gm, s_red), tgc = block_sum(sum_gc, s_red);
    if (tid == 0) {
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

// strips and row segments for an H x W image.  A strip of kFT threads yields kFT - 20 output columns: the launch picks the
// instantiation (64, 128, 192 or 256 threads) that covers the image width with the fewest thread-columns (512 wide: 3 strips
// of 171 columns on 192 threads; ties go to the wider strip).  Segments of 256 rows from 128 M values per batch on (24 x 2048^2:
// 2.25 against 2.32 ms with 128), 128 from 16 M, 64 from 2 M, 16 below (one view of Topo4D's 512 x 375 images: more, shorter workgroups of a single wave - every segment pays 21
// warm-up rows, but a lone view is latency-bound).  T4D_PH_THREADS / T4D_PH_ROWS override (sweeps: tools/sweep_loss_kernels.py).
// (threads = 0: a launch whose tiles are all resident at once takes the TILE kernel - th = 64 when there is at most one tile of
// 64 x 44 pixels per CU (one view of Topo4D's 512 x 375 images: 216), else th = 32 up to two tiles of 32 x 44 per CU; three such
// views are 1,296 tiles and take the strips: 49 against 60 us)
static int photo_cus()
{
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
        cus = n;
    }
    return cus;
}

static void photo_tiling(int32_t n_views, int32_t H, int32_t W, int *tx, int *ty, int *tw, int *th, int *threads)
{
    const long long work = (long long)n_views * 3 * H * W;
    const char *force = getenv("T4D_PH_TILE");              // 1 / 0: force the tile / the strip kernel (tests, sweeps)
    const long long t64 = (long long)((W + 43) / 44) * ((H + 63) / 64) * n_views * 3;
    const long long t32 = (long long)((W + 43) / 44) * ((H + 31) / 32) * n_views * 3;
    if (force ? atoi(force) != 0 : t32 <= 2ll * photo_cus()) {
        *threads = 0; *tw = 44;
        *th = (atoi(force ? force : "0") == 32 || t64 > photo_cus()) ? 32 : 64;        // (T4D_PH_TILE=32 forces the small shape)
        *tx = (W + *tw - 1) / *tw; *ty = (H + *th - 1) / *th;
        return;
    }
    int best = 0;
    long long best_cost = 0;
    for (int ft = 64; ft <= 256; ft += 64) {
        const int strips = (W + (ft - 4 * kR) - 1) / (ft - 4 * kR);
        const long long cost = (long long)strips * ft;
        if (best == 0 || cost <= best_cost) { best = ft; best_cost = cost; }
    }
    if (work < (1ll << 20)) best = 64;
    if (const char *e = getenv("T4D_PH_THREADS")) { const int ft = atoi(e); if (ft == 64 || ft == 128 || ft == 192 || ft == 256) best = ft; }
    *threads = best;
    *tx = (W + (best - 4 * kR) - 1) / (best - 4 * kR);
    *tw = (W + *tx - 1) / *tx;
    *th = work >= (1ll << 27) ? 256 : (work >= (1ll << 24) ? 128 : (work >= (1ll << 21) ? 64 : 16));
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
    if ((size_t)H * W > ((size_t)1 << 30)) return t4d_internal_fail(T4D_ERR_ARG, "t4d_photometric_loss: more than 2^30 pixels per plane%s", "");
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
    if (ft == 0 && P.th == 64) hipLaunchKernelGGL((k_photo_tile<64, 1024>), grid, dim3(1024), 0, stream, P);
    else if (ft == 0) hipLaunchKernelGGL((k_photo_tile<32, 512>), grid, dim3(512), 0, stream, P);
    else if (ft == 64) hipLaunchKernelGGL(k_photo_stream<64>, grid, dim3(64), 0, stream, P);
    else if (ft == 128) hipLaunchKernelGGL(k_photo_stream<128>, grid, dim3(128), 0, stream, P);
    else if (ft == 192) hipLaunchKernelGGL(k_photo_stream<192>, grid, dim3(192), 0, stream, P);
    else hipLaunchKernelGGL(k_photo_stream<256>, grid, dim3(256), 0, stream, P);
    hipLaunchKernelGGL(k_photo_final, dim3(n_views), dim3(kBlock), 0, stream, P);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return t4d_internal_fail(T4D_ERR_HIP, "t4d_photometric_loss launch: %s", hipGetErrorString(e));
    return T4D_OK;
}
