// t4d_targets.hip — the two small pieces of Topo4D's loss assembly that sit between the dataset and the photometric loss in
// the branches train.py actually executes (train.py:631-632: use_mask = True, use_mask_dense = False):
//
//  * t4d_label_mask_target: helpers.get_mask (helpers.py:811-823) on the face-parsing label image of a camera, and the masked
//    target of get_loss's later-frame branch (train.py:320-326: masked_gt = gt.clone(); masked_gt[filtered_mask == 1] *= 0.1).
//    The reference recomputes both in EVERY iteration (five torch ops over [3,H,W], 1,100 iterations per frame) although they
//    depend on (frame, camera) only; here they are one launch per frame for all its cameras, and the iteration reads the result.
//  * t4d_soft_color_loss: helpers.l1_loss_v2 (helpers.py:119-120) between dense_rgb_colors and dense_init_colors, the
//    'soft_color' term of get_loss_dense (train.py:407, weight 0.02 at train.py:541-543), with its gradient ADDED to the
//    rasterizer's dL/dcolours so that the texture iteration stays a hand-chained sequence of library calls.
//
// Everything here is streaming, HBM-bound work: 16-byte accesses where the layout allows, one pass.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/topo4d_raster.h"

#define T4D_EXPORT extern "C" __attribute__((visibility("default")))
int t4d_internal_fail(int code, const char *fmt, const char *a);

namespace {

constexpr int kBlock = 256;

struct LabelArgs {
    float color[T4D_MAX_MASK_LABELS][3];
    int n;
};

// One thread per pixel of one view: the three channels of a pixel decide together (torch.all(..., dim=0), helpers.py:819) and
// the decision is written to all three (torch.tile(..., (3,1,1)), :820).
__global__ __launch_bounds__(kBlock) void k_label_mask_target(const float *mask_image, const LabelArgs L, const float *gt,
                                                              const float scale, const size_t plane, float *filtered, float *target)
{
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= plane) return;
    const size_t base = (size_t)blockIdx.y * 3 * plane + i;
    // `mask = mask * 255` (helpers.py:814) and `mask - target_color` (:819) are two float32 roundings in the reference: a fused
    // multiply-add would round once and can land on the other side of `< 1`
    const float m0 = __fmul_rn(mask_image[base], 255.f);
    const float m1 = __fmul_rn(mask_image[base + plane], 255.f);
    const float m2 = __fmul_rn(mask_image[base + 2 * plane], 255.f);
    bool hit = false;
    for (int k = 0; k < L.n; k++) {
        const bool c = fabsf(__fsub_rn(m0, L.color[k][0])) < 1.f && fabsf(__fsub_rn(m1, L.color[k][1])) < 1.f &&
                       fabsf(__fsub_rn(m2, L.color[k][2])) < 1.f;
        hit = hit || c;
    }
    if (filtered) {
        const float f = hit ? 1.f : 0.f;
        filtered[base] = f; filtered[base + plane] = f; filtered[base + 2 * plane] = f;
    }
    if (target) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float g = gt[base + c * plane];
            target[base + c * plane] = hit ? __fmul_rn(g, scale) : g;
        }
    }
}

// ---- soft colour ----------------------------------------------------------------------------------------------------
constexpr int kScBlocks = 1024;      // partial sums of the loss: fixed count, fixed order => bit-reproducible

__global__ __launch_bounds__(kBlock) void k_soft_color(const float *x, const float *y, const size_t n, const float gw,
                                                       float *grad, const int accumulate, float *partial)
{
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        const float d = x[i] - y[i];
        s += fabsf(d);
        if (grad) {
            // torch: abs' = sign (0 at 0), times the incoming weight / rows (mean over rows, sum over the row: helpers.py:120)
            const float g = d > 0.f ? gw : (d < 0.f ? -gw : (d == d ? 0.f : d));
            grad[i] = accumulate ? grad[i] + g : g;
        }
    }
    // fixed-order reduction: lanes of a wave by xor butterflies, the four waves through LDS
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    __shared__ float w[kBlock / 64];
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (w[0] + w[1]) + (w[2] + w[3]);
}

__global__ __launch_bounds__(kScBlocks) void k_soft_color_final(const float *partial, const int n_partial, const float rows, float *loss)
{
    float s = threadIdx.x < n_partial ? partial[threadIdx.x] : 0.f;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    __shared__ float w[kScBlocks / 64];
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int k = 0; k < kScBlocks / 64; k++) t += w[k];
        *loss = t / rows;
    }
}

}  // namespace

T4D_EXPORT int t4d_label_mask_target(int32_t n_views, int32_t H, int32_t W, const float *mask_image, const float *label_colors,
                                     int32_t n_labels, const float *gt, float scale, float *filtered_mask, float *target,
                                     void *hip_stream)
{
    if (n_views < 1 || H < 1 || W < 1 || !mask_image || !label_colors || n_labels < 0 || n_labels > T4D_MAX_MASK_LABELS ||
        (target && !gt) || (!filtered_mask && !target))
        return t4d_internal_fail(T4D_ERR_ARG, "t4d_label_mask_target: bad arguments%s", "");
    if (n_views > 65535) return t4d_internal_fail(T4D_ERR_ARG, "t4d_label_mask_target: too many views%s", "");
    LabelArgs L;
    memset(&L, 0, sizeof(L));
    L.n = n_labels;
    for (int k = 0; k < n_labels; k++)
        for (int c = 0; c < 3; c++) L.color[k][c] = label_colors[3 * k + c];
    const size_t plane = (size_t)H * W;
    const size_t blocks = (plane + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffull) return t4d_internal_fail(T4D_ERR_ARG, "t4d_label_mask_target: image too large%s", "");
    hipLaunchKernelGGL(k_label_mask_target, dim3((unsigned)blocks, n_views), dim3(kBlock), 0, (hipStream_t)hip_stream, mask_image, L, gt,
                       scale, plane, filtered_mask, target);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return t4d_internal_fail(T4D_ERR_HIP, "t4d_label_mask_target launch: %s", hipGetErrorString(e));
    return T4D_OK;
}

T4D_EXPORT size_t t4d_soft_color_scratch_bytes(void) { return kScBlocks * sizeof(float); }

T4D_EXPORT int t4d_soft_color_loss(int64_t rows, int32_t width, const float *x, const float *y, float weight, float *loss,
                                   float *grad, int32_t accumulate, void *scratch, size_t scratch_bytes, void *hip_stream)
{
    if (rows < 1 || width < 1 || !x || !y || !loss || !scratch)
        return t4d_internal_fail(T4D_ERR_ARG, "t4d_soft_color_loss: bad arguments%s", "");
    if (scratch_bytes < t4d_soft_color_scratch_bytes())
        return t4d_internal_fail(T4D_ERR_STATE_SIZE, "t4d_soft_color_loss: scratch too small%s", "");
    const size_t n = (size_t)rows * width;
    size_t blocks = (n + kBlock - 1) / kBlock;
    if (blocks > kScBlocks) blocks = kScBlocks;
    // d(weight * mean_rows(sum_width |x - y|)) / dx = (weight / rows) * sign(x - y): the division is torch's, in float32
    const float gw = weight / (float)rows;
    hipStream_t stream = (hipStream_t)hip_stream;
    hipLaunchKernelGGL(k_soft_color, dim3((unsigned)blocks), dim3(kBlock), 0, stream, x, y, n, gw, grad, (int)accumulate, (float *)scratch);
    hipLaunchKernelGGL(k_soft_color_final, dim3(1), dim3(kScBlocks), 0, stream, (const float *)scratch, (int)blocks, (float)rows, loss);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return t4d_internal_fail(T4D_ERR_HIP, "t4d_soft_color_loss launch: %s", hipGetErrorString(e));
    return T4D_OK;
}
