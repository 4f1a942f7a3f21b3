// t4d_activations.h - Topo4D's parameter activations (reference helpers.py:95-97: rotations = F.normalize(unnorm_rotations),
// opacities = sigmoid(logit_opacities), scales = exp(log_scales)) and their vector-Jacobian products: ONE definition, included by
// t4d_optim.hip (t4d_activate_forward / t4d_activate_backward) and by t4d_raster.hip (T4D_FLAG_RAW_PARAMS: the rasterizer takes
// the optimiser's raw parameters and returns gradients with respect to them), so that both routes round identically.
// F.normalize(x, p=2, dim=1, eps=1e-12) = x / max(||x||_2, eps).
#pragma once

constexpr float kT4DNormEps = 1e-12f;

__device__ __forceinline__ float4 t4d_act_normalize(const float4 q)
{
#pragma clang fp contract(off)
    const float n = fmaxf(sqrtf(fmaf(q.w, q.w, fmaf(q.z, q.z, fmaf(q.y, q.y, q.x * q.x)))), kT4DNormEps);
    return make_float4(q.x / n, q.y / n, q.z / n, q.w / n);
}
__device__ __forceinline__ float t4d_act_sigmoid(const float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float t4d_act_exp(const float x) { return expf(x); }

// y = x / n:  dx = (g - y (y . g)) / n;  clamped denominator (||x|| <= eps): y = x / eps
__device__ __forceinline__ float4 t4d_act_normalize_bwd(const float4 q, const float4 g)
{
#pragma clang fp contract(off)
    const float nn = sqrtf(fmaf(q.w, q.w, fmaf(q.z, q.z, fmaf(q.y, q.y, q.x * q.x))));
    if (nn > kT4DNormEps) {
        const float inv = 1.0f / nn;
        const float4 y = make_float4(q.x * inv, q.y * inv, q.z * inv, q.w * inv);
        const float yg = fmaf(y.w, g.w, fmaf(y.z, g.z, fmaf(y.y, g.y, y.x * g.x)));
        return make_float4((g.x - y.x * yg) * inv, (g.y - y.y * yg) * inv, (g.z - y.z * yg) * inv, (g.w - y.w * yg) * inv);
    }
    return make_float4(g.x / kT4DNormEps, g.y / kT4DNormEps, g.z / kT4DNormEps, g.w / kT4DNormEps);
}
// s = sigmoid(x): dx = g s (1 - s);   e = exp(x): dx = g e
__device__ __forceinline__ float t4d_act_sigmoid_bwd(const float s, const float g)
{
#pragma clang fp contract(off)
    return g * s * (1.0f - s);
}
__device__ __forceinline__ float t4d_act_exp_bwd(const float e, const float g) { return g * e; }
