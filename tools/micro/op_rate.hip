// micro-benchmark: issue cost of individual VALU instruction kinds on MI355X (8 independent registers per wave, 8 waves/SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 0.01f + i;
    const float m = 1.0001f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (MODE == 0) {
#define S(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(m));
                REP8(S)
#undef S
            } else if (MODE == 1) {
#define S(i) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
                REP8(S)
#undef S
            } else if (MODE == 2) {
#define S(i) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
                REP8(S)
#undef S
            } else if (MODE == 3) {
#define S(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                REP8(S)
#undef S
            } else if (MODE == 4) {
#define S(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double *>(&a[i & 6])) : "v"(*reinterpret_cast<const double *>(&a[(i & 6) ^ 2])));
                REP8(S)
#undef S
            } else if (MODE == 5) {
#define S(i) asm volatile("v_cmp_lt_f32 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m) : "vcc");
                REP8(S)
#undef S
            } else if (MODE == 6) {
#define S(i) asm volatile("v_add_f32_dpp %0, %0, %0 row_shl:4 row_mask:0xf bank_mask:0x5" : "+v"(a[i]));
                REP8(S)
#undef S
            } else if (MODE == 7) {
#define S(i) asm volatile("v_mov_b32_dpp %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
                REP8(S)
#undef S
            }
        }
    }
    float s = 0; for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char *name, int per8)
{
    float *out; (void)hipMalloc(&out, 1 << 24);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int iters = 2000, blocks = 256 * 8;
    k<MODE><<<blocks, 256>>>(out, iters); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); k<MODE><<<blocks, 256>>>(out, iters); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double n_instr = (double)iters * 4 * per8 * 8;          // per wave ... 8 waves per SIMD
    printf("%-34s %.2f SIMD-cycles per instruction (2.4 GHz)\n", name, ms * 1e-3 * 2.4e9 / n_instr);
    (void)hipFree(out);
}
int main()
{
    run<0>("v_fma_f32", 8); run<1>("v_add_f32_dpp row_ror:8", 8); run<2>("v_add_f32_dpp quad_perm", 8); run<3>("v_exp_f32", 8);
    run<4>("v_pk_mul_f32", 8); run<5>("v_cmp + v_cndmask (pair)", 8); run<6>("v_add_f32_dpp row_shl:4 bank 0x5", 8);
    run<7>("v_mov_b32_dpp row_ror:8", 8);
    return 0;
}
