// t4d_optim.hip — fused multi-tensor Adam step + region pins for Topo4D's per-view optimisation loop
// (SURVEY.md §8f rank 3).
//
// One launch replaces `optimizer.step()` (torch.optim.Adam with one parameter group per tensor, eps = 1e-15,
// reference train.py:272-297, :672) AND the ~16 masked assignments that follow it in every iteration
// (`params[name][region_mask] = frozen_values`, train.py:676-700): each tensor may carry a per-row pin mask and the
// values its pinned rows must hold after the step.  Arithmetic follows torch.optim.Adam (no weight decay, no amsgrad):
//     m <- m + (1-b1)(g - m);  v <- b2 v + (1-b2) g^2;  p <- p - (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// with the bias corrections evaluated on the host in double precision, as torch does.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../../include/topo4d_raster.h"
#include "t4d_activations.h"

#define T4D_EXPORT extern "C" __attribute__((visibility("default")))
int t4d_internal_fail(int code, const char *fmt, const char *a);

namespace {

constexpr int kMaxTensors = T4D_ADAM_MAX_TENSORS;
constexpr int kBlock = 256;

struct AdamArgs {
    T4DAdamTensor t[kMaxTensors];
    long long first_block[kMaxTensors + 1];    // exclusive prefix of the per-tensor block counts
    float step_size[kMaxTensors];              // lr / (1 - beta1^t)                (host-side hyper-parameters)
    float inv_bc2_sqrt[kMaxTensors];           // 1 / sqrt(1 - beta2^t)
    int32_t *step_dev;                         // device-side hyper-parameters (graph replays): one step counter PER WORKGROUP of the launch
    const float *lr_dev;
    int n;
    float beta1, beta2, eps;
};

template <bool DEV>
__global__ __launch_bounds__(kBlock) void k_adam_pin(const AdamArgs A)
{
#pragma clang fp contract(off)      // only the two explicit fmaf below fuse (torch's lerp / addcmul forms); nothing else may
    int k = 0;
#pragma unroll
    for (int i = 1; i < kMaxTensors; i++) k += (i < A.n && (long long)blockIdx.x >= A.first_block[i]) ? 1 : 0;
    const T4DAdamTensor &T = A.t[k];
    float step_size = A.step_size[k], inv_bc2_sqrt = A.inv_bc2_sqrt[k];
    if (DEV) {
        // same double-precision bias corrections as the host path, from the counters kept in device memory
        __shared__ float s_hyper[2];
        if (threadIdx.x == 0) {
            // Every workgroup keeps its OWN copy of its tensor's step count and advances it itself: no tick launch in front of the
            // step (a kernel that does nothing still lasts 4.5 us between its neighbours in a replayed graph), no race - nobody
            // else touches this word - and all copies of a tensor agree because they all count the same launches.
            int32_t st_i = A.step_dev[blockIdx.x];
            if (T.grad) { st_i += 1; A.step_dev[blockIdx.x] = st_i; }
            const double st = T.grad ? (double)st_i : 1.0;
            const double bc1 = 1.0 - pow((double)A.beta1, st), bc2 = 1.0 - pow((double)A.beta2, st);
            s_hyper[0] = (float)((double)A.lr_dev[k] / bc1);
            s_hyper[1] = (float)(1.0 / sqrt(bc2));
        }
        __syncthreads();
        step_size = s_hyper[0]; inv_bc2_sqrt = s_hyper[1];
    }
    const long long i = ((long long)blockIdx.x - A.first_block[k]) * kBlock + threadIdx.x;
    const long long numel = T.rows * T.width;
    if (i >= numel) return;
    float p = T.param[i];
    if (T.grad) {
        const float g = T.grad[i];
        float m = T.exp_avg[i], v = T.exp_avg_sq[i];
        m = fmaf(1.f - A.beta1, g - m, m);
        v = fmaf(1.f - A.beta2, g * g, A.beta2 * v);
        T.exp_avg[i] = m;
        T.exp_avg_sq[i] = v;
        const float denom = sqrtf(v) * inv_bc2_sqrt + A.eps;
        p = p - step_size * (m / denom);
        if (T.flags & T4D_ADAM_CLEAR_GRAD) const_cast<float *>(T.grad)[i] = 0.f;
    }
    if (T.pin_mask && T.pin_mask[i / T.width]) p = T.pin_values[i];
    T.param[i] = p;
}

int adam_launch(const T4DAdamTensor *tensors, int32_t n_tensors, float beta1, float beta2, float eps, int32_t *step_dev,
                const float *lr_dev, void *hip_stream)
{
    if (!tensors || n_tensors < 1 || n_tensors > kMaxTensors)
        return t4d_internal_fail(T4D_ERR_ARG, "t4d_adam_pin_step: 1..T4D_ADAM_MAX_TENSORS tensors%s", "");
    if ((step_dev == nullptr) != (lr_dev == nullptr))
        return t4d_internal_fail(T4D_ERR_ARG, "t4d_adam_pin_step_graph: step_dev and lr_dev go together%s", "");
    AdamArgs A;
    memset(&A, 0, sizeof(A));
    A.n = n_tensors; A.beta1 = beta1; A.beta2 = beta2; A.eps = eps;
    A.step_dev = step_dev; A.lr_dev = lr_dev;
    long long blocks = 0;
    for (int k = 0; k < n_tensors; k++) {
        const T4DAdamTensor &t = tensors[k];
        if (!t.param || t.rows < 0 || t.width < 1 || (t.grad && (!t.exp_avg || !t.exp_avg_sq)) || ((t.pin_mask == nullptr) != (t.pin_values == nullptr)))
            return t4d_internal_fail(T4D_ERR_ARG, "t4d_adam_pin_step: inconsistent tensor descriptor%s", "");
        if (!step_dev && t.grad && t.step < 1) return t4d_internal_fail(T4D_ERR_ARG, "t4d_adam_pin_step: step must be >= 1%s", "");
        A.t[k] = t;
        const double st = t.grad ? (double)t.step : 1.0;
        const double bc1 = 1.0 - pow((double)beta1, st), bc2 = 1.0 - pow((double)beta2, st);
        A.step_size[k] = (float)((double)t.lr / bc1);
        A.inv_bc2_sqrt[k] = (float)(1.0 / sqrt(bc2));
        A.first_block[k] = blocks;
        blocks += (t.rows * t.width + kBlock - 1) / kBlock;
    }
    A.first_block[n_tensors] = blocks;
    if (blocks == 0) return T4D_OK;
    if (blocks > 0x7fffffffLL) return t4d_internal_fail(T4D_ERR_ARG, "t4d_adam_pin_step: too many elements%s", "");
    if (step_dev) {
        hipLaunchKernelGGL(k_adam_pin<true>, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)hip_stream, A);
    } else {
        hipLaunchKernelGGL(k_adam_pin<false>, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)hip_stream, A);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return t4d_internal_fail(T4D_ERR_HIP, "t4d_adam_pin_step launch: %s", hipGetErrorString(e));
    return T4D_OK;
}

}  // namespace

T4D_EXPORT int t4d_adam_pin_step(const T4DAdamTensor *tensors, int32_t n_tensors, float beta1, float beta2, float eps,
                                 void *hip_stream)
{
    return adam_launch(tensors, n_tensors, beta1, beta2, eps, nullptr, nullptr, hip_stream);
}

T4D_EXPORT int64_t t4d_adam_step_counters(const T4DAdamTensor *tensors, int32_t n_tensors)
{
    if (!tensors || n_tensors < 1 || n_tensors > kMaxTensors) return 0;
    long long blocks = 0;
    for (int k = 0; k < n_tensors; k++) blocks += (tensors[k].rows * tensors[k].width + kBlock - 1) / kBlock;
    return (int64_t)blocks;
}

T4D_EXPORT int t4d_adam_pin_step_graph(const T4DAdamTensor *tensors, int32_t n_tensors, float beta1, float beta2, float eps,
                                       int32_t *step_dev, int64_t n_step_counters, const float *lr_dev, void *hip_stream)
{
    if (!step_dev || !lr_dev) return t4d_internal_fail(T4D_ERR_ARG, "t4d_adam_pin_step_graph: step_dev and lr_dev are required%s", "");
    // the launch writes one counter per workgroup: an array sized for other shapes (a tensor replaced by a larger one) would be
    // written out of bounds, and every later tensor would read another tensor's counters
    if (n_step_counters != t4d_adam_step_counters(tensors, n_tensors))
        return t4d_internal_fail(T4D_ERR_ARG, "t4d_adam_pin_step_graph: n_step_counters does not match t4d_adam_step_counters() of these tensors%s", "");
    return adam_launch(tensors, n_tensors, beta1, beta2, eps, step_dev, lr_dev, hip_stream);
}

// ---------------------------------------------------------------------------------------------------------
// Dense-attribute interpolation: restates reference helpers.py:237-253
// `compute_vertex_attribute_by_weight_2` — every UV-densified vertex is the weighted sum of the four corners of its
// "father" quad — which Topo4D runs in numpy on the host once per frame after a device->host copy (train.py:504-506).
// The reference accumulates in float64 (np.zeros / np.sum defaults) and the result is cast to float32 afterwards
// (train.py:259-261 `.float()`); the kernel does the same, so the float32 output is bit-identical.
// ---------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void k_dense_interp(const float *attr, const int32_t *quad_faces, const int32_t *father,
                                                      const double *weight, long long n_coarse, long long n_dense, int width,
                                                      float *out)
{
#pragma clang fp contract(off)      // numpy rounds every product and every partial sum: no fused multiply-adds here
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (n_coarse + n_dense) * width;
    if (i >= total) return;
    const long long row = i / width;
    const int c = (int)(i - row * width);
    if (row < n_coarse) { out[i] = attr[i]; return; }
    const long long d = row - n_coarse;
    const int32_t *q = quad_faces + 4 * (long long)father[d];
    const double *w = weight + 4 * d;
    double s = (double)attr[(long long)q[0] * width + c] * w[0];            // np.sum over 4 terms: left to right
    s += (double)attr[(long long)q[1] * width + c] * w[1];
    s += (double)attr[(long long)q[2] * width + c] * w[2];
    s += (double)attr[(long long)q[3] * width + c] * w[3];
    out[i] = (float)s;
}
}  // namespace

T4D_EXPORT int t4d_dense_interpolate(const float *attribute, const int32_t *quad_faces, const int32_t *vertex_father,
                                     const double *weight, int64_t n_coarse, int64_t n_dense, int32_t width, float *out,
                                     void *hip_stream)
{
    if (!attribute || !out || n_coarse < 0 || n_dense < 0 || width < 1 || (n_dense > 0 && (!quad_faces || !vertex_father || !weight)))
        return t4d_internal_fail(T4D_ERR_ARG, "t4d_dense_interpolate: bad arguments%s", "");
    const long long total = (n_coarse + n_dense) * width;
    if (total == 0) return T4D_OK;
    hipLaunchKernelGGL(k_dense_interp, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, attribute,
                       quad_faces, vertex_father, weight, (long long)n_coarse, (long long)n_dense, (int)width, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return t4d_internal_fail(T4D_ERR_HIP, "t4d_dense_interpolate launch: %s", hipGetErrorString(e));
    return T4D_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Parameter activations of params2rendervar (helpers.py:91-100): rotations = F.normalize(unnorm_rotations),
// opacities = sigmoid(logit_opacities), scales = exp(log_scales) - three tiny torch kernels forward and, with autograd,
// about a dozen backward, every iteration.  One launch each way here.
// F.normalize(x, p=2, dim=1, eps=1e-12) = x / max(||x||_2, eps).
// ---------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void k_activate_fwd(long long P, const float4 *unnorm_rot, const float *logit_op, const float *log_scale,
                                                      float4 *rot, float *op, float *scale)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    rot[i] = t4d_act_normalize(unnorm_rot[i]);
    op[i] = t4d_act_sigmoid(logit_op[i]);
#pragma unroll
    for (int k = 0; k < 3; k++) scale[3 * i + k] = t4d_act_exp(log_scale[3 * i + k]);
}

__global__ __launch_bounds__(256) void k_activate_bwd(long long P, const float4 *unnorm_rot, const float *op, const float *scale,
                                                      const float4 *g_rot, const float *g_op, const float *g_scale,
                                                      float4 *g_unnorm, float *g_logit, float *g_log_scale)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    if (g_unnorm) g_unnorm[i] = t4d_act_normalize_bwd(unnorm_rot[i], g_rot ? g_rot[i] : make_float4(0.f, 0.f, 0.f, 0.f));
    if (g_logit) g_logit[i] = g_op ? t4d_act_sigmoid_bwd(op[i], g_op[i]) : 0.f;
    if (g_log_scale) {
#pragma unroll
        for (int k = 0; k < 3; k++) g_log_scale[3 * i + k] = g_scale ? t4d_act_exp_bwd(scale[3 * i + k], g_scale[3 * i + k]) : 0.f;
    }
}
}  // namespace

T4D_EXPORT int t4d_activate_forward(int64_t P, const float *unnorm_rotations, const float *logit_opacities, const float *log_scales,
                                    float *rotations, float *opacities, float *scales, void *hip_stream)
{
    if (P < 0 || (P > 0 && (!unnorm_rotations || !logit_opacities || !log_scales || !rotations || !opacities || !scales)))
        return t4d_internal_fail(T4D_ERR_ARG, "t4d_activate_forward: bad arguments%s", "");
    if (P == 0) return T4D_OK;
    hipLaunchKernelGGL(k_activate_fwd, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, (long long)P,
                       reinterpret_cast<const float4 *>(unnorm_rotations), logit_opacities, log_scales,
                       reinterpret_cast<float4 *>(rotations), opacities, scales);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return t4d_internal_fail(T4D_ERR_HIP, "t4d_activate_forward launch: %s", hipGetErrorString(e));
    return T4D_OK;
}

T4D_EXPORT int t4d_activate_backward(int64_t P, const float *unnorm_rotations, const float *opacities, const float *scales,
                                     const float *dL_drotations, const float *dL_dopacities, const float *dL_dscales,
                                     float *dL_dunnorm_rotations, float *dL_dlogit_opacities, float *dL_dlog_scales, void *hip_stream)
{
    // (a NULL cotangent counts as zeros, a NULL output is not wanted: only the forward's inputs / outputs are required)
    if (P < 0 || (P > 0 && (!unnorm_rotations || !opacities || !scales)))
        return t4d_internal_fail(T4D_ERR_ARG, "t4d_activate_backward: bad arguments%s", "");
    if (P == 0) return T4D_OK;
    hipLaunchKernelGGL(k_activate_bwd, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, (long long)P,
                       reinterpret_cast<const float4 *>(unnorm_rotations), opacities, scales,
                       reinterpret_cast<const float4 *>(dL_drotations), dL_dopacities, dL_dscales,
                       reinterpret_cast<float4 *>(dL_dunnorm_rotations), dL_dlogit_opacities, dL_dlog_scales);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return t4d_internal_fail(T4D_ERR_HIP, "t4d_activate_backward launch: %s", hipGetErrorString(e));
    return T4D_OK;
}


// ---------------------------------------------------------------------------------------------------------
// view-summed gradients (include/topo4d_raster.h: t4d_sum_views)
// ---------------------------------------------------------------------------------------------------------
namespace {
struct SumArgs {
    const float *src[T4D_SUM_MAX_TENSORS];
    float *dst[T4D_SUM_MAX_TENSORS];
    long long n[T4D_SUM_MAX_TENSORS];
    long long first_block[T4D_SUM_MAX_TENSORS + 1];
    int V, count;
};

__global__ __launch_bounds__(kBlock) void k_sum_views(const SumArgs a)
{
    int k = 0;
#pragma unroll
    for (int i = 1; i < T4D_SUM_MAX_TENSORS; i++)
        if (i < a.count && (long long)blockIdx.x >= a.first_block[i]) k = i;
    const long long n = a.n[k];
    const long long i4 = (((long long)blockIdx.x - a.first_block[k]) * kBlock + threadIdx.x) * 4;
    if (i4 >= n) return;
    const float *s = a.src[k];
    if (i4 + 4 <= n && (n & 3) == 0 && (((uintptr_t)s | (uintptr_t)a.dst[k]) & 15) == 0) {
        float4 acc = *reinterpret_cast<const float4 *>(s + i4);
        for (int v = 1; v < a.V; v++) {
            const float4 x = *reinterpret_cast<const float4 *>(s + (long long)v * n + i4);
            acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
        }
        *reinterpret_cast<float4 *>(a.dst[k] + i4) = acc;
    } else {
        for (long long i = i4; i < min(i4 + 4, n); i++) {
            float acc = s[i];
            for (int v = 1; v < a.V; v++) acc += s[(long long)v * n + i];
            a.dst[k][i] = acc;
        }
    }
}
}  // namespace

T4D_EXPORT int t4d_sum_views(int32_t n_views, int32_t n_tensors, const float *const *src, float *const *dst, const int64_t *n_per_view,
                             void *hip_stream)
{
    if (n_views < 1 || n_tensors < 1 || n_tensors > T4D_SUM_MAX_TENSORS || !src || !dst || !n_per_view)
        return t4d_internal_fail(T4D_ERR_ARG, "t4d_sum_views: bad arguments%s", "");
    SumArgs a;
    memset(&a, 0, sizeof(a));
    a.V = n_views;
    long long blocks = 0;
    for (int k = 0; k < n_tensors; k++) {
        if (!src[k] || !dst[k] || n_per_view[k] <= 0) continue;
        a.src[a.count] = src[k]; a.dst[a.count] = dst[k]; a.n[a.count] = n_per_view[k];
        a.first_block[a.count] = blocks;
        blocks += (n_per_view[k] + 4 * kBlock - 1) / (4 * kBlock);
        a.count++;
    }
    a.first_block[a.count] = blocks;
    if (a.count == 0) return T4D_OK;
    if (blocks > 0x7fffffffLL) return t4d_internal_fail(T4D_ERR_ARG, "t4d_sum_views: too many elements%s", "");
    hipLaunchKernelGGL(k_sum_views, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)hip_stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return t4d_internal_fail(T4D_ERR_HIP, "t4d_sum_views launch: %s", hipGetErrorString(e));
    return T4D_OK;
}
