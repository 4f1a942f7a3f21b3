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

#include "../../include/topo4d_raster.h"

#define T4D_EXPORT extern "C" __attribute__((visibility("default")))
int t4d_internal_fail(int code, const char *fmt, const char *a);

namespace {

// One workgroup = a 64 x 16 pixel tile of one channel of one view, 256 threads.  The separable 11-tap window runs as a
// vertical pass (each task: one column, FOUR consecutive output rows, 14 input rows read once into registers) followed by
// a horizontal pass (each thread: one row, FOUR consecutive output columns) — a register sliding window, so an output costs
// ~3.5 + 7 LDS reads instead of 11 + 11 x (number of filtered maps).
#ifndef T4D_PH_TW
#define T4D_PH_TW 64
#endif
constexpr int kTW = T4D_PH_TW, kTH = 16;  // tile (kTW a multiple of 4)
constexpr int kR = 5;              // window radius (11 taps)
constexpr int kIW = kTW + 2 * kR;  // 74
constexpr int kIH = kTH + 2 * kR;  // 26
constexpr int kPitch = (kIW + 2) | 1;         // row pitch (elements) of the vertically filtered maps: odd multiple keeps the
                                   // lane->row mapping of the horizontal pass free of LDS bank conflicts
constexpr int kBlock = 256;
constexpr float kC1 = 0.01f * 0.01f, kC2 = 0.03f * 0.03f;

struct PhP {
    int V, H, W, tx, ty;
    const float *im, *gt, *cam_m, *cam_c, *weight;
    float *loss, *dL_dim, *dL_dm, *dL_dc;
    float4 *D;           // [V*3*H*W] adjoint maps (D1, D2, D3, -)
    float *part_loss;    // [V*3*tiles][2]  (sum |x'-y|, sum S)
    float *part_cam;     // [V*3*tiles][2]  (sum g'*(x'-c), sum g')
    float win[11];
};

__device__ __forceinline__ float block_sum(float v, float *s_red)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    const int tid = threadIdx.x;
    __syncthreads();
    if ((tid & 63) == 0) s_red[tid >> 6] = v;
    __syncthreads();
    return (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

__global__ __launch_bounds__(kBlock) void k_photo_stats(const PhP P)
{
    __shared__ float2 s_in[kIH][kIW + 1];                // (x', gt) with halo, zero padded
    __shared__ float4 s_v4[kTH][kPitch];                 // vertically filtered (x, y, x^2, y^2)
    __shared__ float s_v1[kTH][kPitch];                  // vertically filtered x*y
    __shared__ float s_red[4];
    const int tid = threadIdx.x;
    const int vc = blockIdx.z, v = vc / 3;
    const int x0 = blockIdx.x * kTW, y0 = blockIdx.y * kTH;
    const size_t HW = (size_t)P.H * P.W;
    const float *im = P.im + (size_t)vc * HW, *gt = P.gt + (size_t)vc * HW;
    const float em = P.cam_m ? expf(P.cam_m[vc]) : 1.f, cc = P.cam_c ? P.cam_c[vc] : 0.f;
    for (int i = tid; i < kIH * kIW; i += kBlock) {
        const int r = i / kIW, c = i - r * kIW;
        const int yy = y0 + r - kR, xx = x0 + c - kR;
        float a = 0.f, b = 0.f;                                  // zero padding (external.py:86 padding=5)
        if (yy >= 0 && yy < P.H && xx >= 0 && xx < P.W) {
            a = em * im[(size_t)yy * P.W + xx] + cc;
            b = gt[(size_t)yy * P.W + xx];
        }
        s_in[r][c] = make_float2(a, b);
    }
    __syncthreads();
    for (int t = tid; t < kIW * (kTH / 4); t += kBlock) {        // vertical pass: column `col`, output rows r0..r0+3
        const int run = t / kIW, col = t - run * kIW, r0 = run * 4;
        float sx[4] = { 0, 0, 0, 0 }, sy[4] = { 0, 0, 0, 0 }, sxx[4] = { 0, 0, 0, 0 }, syy[4] = { 0, 0, 0, 0 }, sxy[4] = { 0, 0, 0, 0 };
#pragma unroll
        for (int k = 0; k < 14; k++) {
            const float2 ab = s_in[r0 + k][col];
            const float a = ab.x, b = ab.y, aa = a * a, bb = b * b, abp = a * b;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int tap = k - j;
                if (tap >= 0 && tap < 11) {
                    const float w = P.win[tap];
                    sx[j] += w * a; sy[j] += w * b; sxx[j] += w * aa; syy[j] += w * bb; sxy[j] += w * abp;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            s_v4[r0 + j][col] = make_float4(sx[j], sy[j], sxx[j], syy[j]);
            s_v1[r0 + j][col] = sxy[j];
        }
    }
    __syncthreads();
    // horizontal pass: lane -> row (fastest), 4 consecutive output columns per thread
    const int row = tid & 15, c0 = (tid >> 4) * 4;
    const bool hrun = c0 < kTW;                          // (kTW / 4) * 16 threads take part in the horizontal pass
    float mu1[4] = { 0, 0, 0, 0 }, mu2[4] = { 0, 0, 0, 0 }, ea[4] = { 0, 0, 0, 0 }, ec[4] = { 0, 0, 0, 0 }, eb[4] = { 0, 0, 0, 0 };
#pragma unroll
    for (int k = 0; k < 14; k++) {
        const float4 h4 = hrun ? s_v4[row][c0 + k] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float h1 = hrun ? s_v1[row][c0 + k] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int tap = k - j;
            if (tap >= 0 && tap < 11) {
                const float w = P.win[tap];
                mu1[j] += w * h4.x; mu2[j] += w * h4.y; ea[j] += w * h4.z; ec[j] += w * h4.w; eb[j] += w * h1;
            }
        }
    }
    const int py = y0 + row;
    const float N = 3.f * (float)HW;
    const float g = -0.2f * (P.weight ? P.weight[v] : 1.f) / N;              // dL/dS
    float l1 = 0.f, ss = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int px = x0 + c0 + j;
        if (hrun && px < P.W && py < P.H) {
            const float mu1s = mu1[j] * mu1[j], mu2s = mu2[j] * mu2[j], mu12 = mu1[j] * mu2[j];
            const float s11 = ea[j] - mu1s, s22 = ec[j] - mu2s, s12 = eb[j] - mu12;
            const float A1 = 2.f * mu12 + kC1, A2 = 2.f * s12 + kC2, B1 = mu1s + mu2s + kC1, B2 = s11 + s22 + kC2;
            const float inv = 1.f / (B1 * B2);
            const float S = A1 * A2 * inv;
            ss += S;
            const float2 ab = s_in[row + kR][c0 + j + kR];
            l1 += fabsf(ab.x - ab.y);
            // S = A1 A2 / (B1 B2) with s11 = a - mu1^2, s12 = b - mu1 mu2 (a, b, c = filtered x^2, xy, y^2)
            const float dS_dmu1 = (2.f * mu2[j] * (A2 - A1)) * inv - S * (2.f * mu1[j] / B1 - 2.f * mu1[j] / B2);
            const float dS_da = -S / B2;
            const float dS_db = 2.f * A1 * inv;
            P.D[(size_t)vc * HW + (size_t)py * P.W + px] = make_float4(g * dS_dmu1, g * dS_da, g * dS_db, 0.f);
        }
    }
    const float tl1 = block_sum(l1, s_red);
    const float tss = block_sum(ss, s_red);
    if (tid == 0) {
        const size_t t = ((size_t)vc * P.ty + blockIdx.y) * P.tx + blockIdx.x;
        P.part_loss[2 * t] = tl1; P.part_loss[2 * t + 1] = tss;
    }
}

__global__ __launch_bounds__(kBlock) void k_photo_grad(const PhP P)
{
    __shared__ float4 s_d[kIH][kIW + 1];                 // (D1, D2, D3, -) with halo
    __shared__ float4 s_v[kTH][kPitch];                  // vertically filtered
    __shared__ float s_red[4];
    const int tid = threadIdx.x;
    const int vc = blockIdx.z, v = vc / 3;
    const int x0 = blockIdx.x * kTW, y0 = blockIdx.y * kTH;
    const size_t HW = (size_t)P.H * P.W;
    const float em = P.cam_m ? expf(P.cam_m[vc]) : 1.f, cc = P.cam_c ? P.cam_c[vc] : 0.f;
    for (int i = tid; i < kIH * kIW; i += kBlock) {
        const int r = i / kIW, c = i - r * kIW;
        const int yy = y0 + r - kR, xx = x0 + c - kR;
        const bool in = yy >= 0 && yy < P.H && xx >= 0 && xx < P.W;
        s_d[r][c] = in ? P.D[(size_t)vc * HW + (size_t)yy * P.W + xx] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    for (int t = tid; t < kIW * (kTH / 4); t += kBlock) {
        const int run = t / kIW, col = t - run * kIW, r0 = run * 4;
        float a0[4] = { 0, 0, 0, 0 }, a1[4] = { 0, 0, 0, 0 }, a2[4] = { 0, 0, 0, 0 };
#pragma unroll
        for (int k = 0; k < 14; k++) {
            const float4 d4 = s_d[r0 + k][col];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int tap = k - j;
                if (tap >= 0 && tap < 11) {
                    const float w = P.win[tap];
                    a0[j] += w * d4.x; a1[j] += w * d4.y; a2[j] += w * d4.z;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) s_v[r0 + j][col] = make_float4(a0[j], a1[j], a2[j], 0.f);
    }
    __syncthreads();
    const int row = tid & 15, c0 = (tid >> 4) * 4;
    const bool hrun = c0 < kTW;
    float q0[4] = { 0, 0, 0, 0 }, q1[4] = { 0, 0, 0, 0 }, q2[4] = { 0, 0, 0, 0 };
#pragma unroll
    for (int k = 0; k < 14; k++) {
        const float4 h4 = hrun ? s_v[row][c0 + k] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int tap = k - j;
            if (tap >= 0 && tap < 11) {
                const float w = P.win[tap];
                q0[j] += w * h4.x; q1[j] += w * h4.y; q2[j] += w * h4.z;
            }
        }
    }
    const int py = y0 + row;
    const float N = 3.f * (float)HW, wv = P.weight ? P.weight[v] : 1.f;
    float gm = 0.f, gc = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int px = x0 + c0 + j;
        if (hrun && px < P.W && py < P.H) {
            const size_t o = (size_t)vc * HW + (size_t)py * P.W + px;
            const float imv = P.im[o], y = P.gt[o];
            const float x = em * imv + cc;
            const float d = x - y;
            const float gl1 = 0.8f * wv / N * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
            const float g = q0[j] + 2.f * x * q1[j] + y * q2[j] + gl1;         // dL/dx'
            P.dL_dim[o] = em * g;
            gm += g * (em * imv);                                               // d x'/d cam_m = exp(cam_m) * im
            gc += g;
        }
    }
    const float tgm = block_sum(gm, s_red);
    const float tgc = block_sum(gc, s_red);
    if (tid == 0) {
        const size_t t = ((size_t)vc * P.ty + blockIdx.y) * P.tx + blockIdx.x;
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

T4D_EXPORT size_t t4d_photometric_scratch_bytes(int32_t n_views, int32_t H, int32_t W)
{
    if (n_views < 1 || H < 1 || W < 1) return 0;
    const size_t n = (size_t)n_views * 3 * H * W;
    const size_t tiles = (size_t)((W + kTW - 1) / kTW) * ((H + kTH - 1) / kTH) * n_views * 3;
    return align_up(n * 16) + 2 * align_up(tiles * 8);
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
    P.tx = (W + kTW - 1) / kTW; P.ty = (H + kTH - 1) / kTH;
    P.im = im; P.gt = gt; P.cam_m = cam_m; P.cam_c = cam_c; P.weight = view_weight;
    P.loss = loss; P.dL_dim = dL_dim; P.dL_dm = dL_dcam_m; P.dL_dc = dL_dcam_c;
    const size_t n = (size_t)n_views * 3 * H * W, tiles = (size_t)P.tx * P.ty * n_views * 3;
    char *sc = (char *)scratch;
    P.D = (float4 *)sc;
    P.part_loss = (float *)(sc + align_up(n * 16));
    P.part_cam = (float *)(sc + align_up(n * 16) + align_up(tiles * 8));
    // the reference's window: exp(-(i-5)^2 / (2*1.5^2)) for i = 0..10, as float32, normalised in float32 (external.py:73-76)
    float g[11], sum = 0.f;
    for (int i = 0; i < 11; i++) { g[i] = (float)exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5)); sum += g[i]; }
    for (int i = 0; i < 11; i++) P.win[i] = g[i] / sum;
    hipStream_t stream = (hipStream_t)hip_stream;
    const dim3 grid(P.tx, P.ty, n_views * 3);
    hipLaunchKernelGGL(k_photo_stats, grid, dim3(kBlock), 0, stream, P);
    hipLaunchKernelGGL(k_photo_grad, grid, dim3(kBlock), 0, stream, P);
    hipLaunchKernelGGL(k_photo_final, dim3(n_views), dim3(kBlock), 0, stream, P);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return t4d_internal_fail(T4D_ERR_HIP, "t4d_photometric_loss launch: %s", hipGetErrorString(e));
    return T4D_OK;
}
