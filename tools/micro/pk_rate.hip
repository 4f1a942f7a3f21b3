// micro-benchmark: issue rate of v_pk_fma_f32 vs v_fma_f32 (MI355X)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const float m = 1.0001f, c = 0.5f;
    const v2f m2 = {m, m}, c2 = {c, c};
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 8; u++) { a0 = fmaf(a0, m, c); a1 = fmaf(a1, m, c); a2 = fmaf(a2, m, c); a3 = fmaf(a3, m, c);
                                          a4 = fmaf(a4, m, c); a5 = fmaf(a5, m, c); a6 = fmaf(a6, m, c); a7 = fmaf(a7, m, c); }
        } else {
#pragma unroll
            for (int u = 0; u < 8; u++) { p0 = __builtin_elementwise_fma(p0, m2, c2); p1 = __builtin_elementwise_fma(p1, m2, c2);
                                          p2 = __builtin_elementwise_fma(p2, m2, c2); p3 = __builtin_elementwise_fma(p3, m2, c2); }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
template <int MODE> void run(const char *name, int per_iter)
{
    float *out; (void)hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int iters = 4000, blocks = 256 * 8;
    k<MODE><<<blocks, 256>>>(out, iters); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); k<MODE><<<blocks, 256>>>(out, iters); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double flops = 2.0 * 64 * (double)per_iter * iters * 4 * blocks;    // per wave: 64 lanes * FMAs
    printf("%-14s %.3f ms  %.1f TFLOP/s\n", name, ms, flops / ms * 1e-9);
    (void)hipFree(out);
}
int main() { run<0>("v_fma_f32", 64); run<1>("v_pk_fma_f32", 64); return 0; }
