// micro-benchmark: VALU issue rate per SIMD vs waves/SIMD and independent chains per wave (MI355X)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CHAINS>
__global__ __launch_bounds__(64) void k(float *out, int iters)
{
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x + i;
    const float m = 1.0001f, c = 0.5f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int ch = 0; ch < CHAINS; ch++) a[ch] = fmaf(a[ch], m, c);
    }
    float s = 0; for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int CHAINS> void run(int waves_per_simd)
{
    float *out; (void)hipMalloc(&out, 1 << 24);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int iters = 4000 * 8 / CHAINS, blocks = 256 * 4 * waves_per_simd;      // 64-thread blocks: one wave each
    k<CHAINS><<<blocks, 64>>>(out, iters); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); k<CHAINS><<<blocks, 64>>>(out, iters); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double instr_per_wave = (double)iters * 8 * CHAINS;
    const double cycles = ms * 1e-3 * 2.4e9;
    printf("chains %d  waves/SIMD %d : %.2f cycles per instruction per wave, %.3f instr/cycle/SIMD\n", CHAINS, waves_per_simd,
           cycles / instr_per_wave, instr_per_wave * waves_per_simd / cycles);
    (void)hipFree(out);
}
int main()
{
    for (int w : {1, 2, 4, 5, 8}) { run<1>(w); run<2>(w); run<4>(w); run<8>(w); }
    return 0;
}
