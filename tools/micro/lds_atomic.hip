// micro-benchmark: cost of ds_add_f32 under different address patterns (MI355X)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    __shared__ float s[4][128 * 10];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, row = lane >> 4, i = lane & 15;
    for (int x = tid; x < 4 * 1280; x += 256) (&s[0][0])[x] = 0.f;
    __syncthreads();
    float v = (float)tid;
    for (int it = 0; it < iters; it++) {
        int j;
        if (MODE == 0) j = (it * 7 + row * 13) & 127;          // 4 rows, 4 different splats
        else if (MODE == 1) j = (it * 7) & 127;                 // 4 rows, same splat (4-way same address)
        else if (MODE == 2) j = (it * 7 + (row >> 1) * 13) & 127; // 2-way
        else j = (it * 7 + row * 13) & 127;
        float *p = &s[wave][j * 10 + (i < 10 ? i : 0)];
        if (MODE <= 2) { if (i < 10) atomicAdd(p, v); }
        else if (MODE == 3) { if (i < 10) *p = v; }             // plain store
        else if (MODE == 4) { if (i < 10) { float o = *p; *p = o + v; } }   // plain RMW
        else if (MODE == 5) { if (lane < 10) atomicAdd(p, v); }             // one row only (10 lanes)
        else if (MODE == 6) { if (lane == 0) atomicAdd(p, v); }             // one lane
        else if (MODE == 7) { if (i < 10) atomicAdd(reinterpret_cast<unsigned int *>(p), (unsigned int)it); }   // integer atomic
        else if (MODE == 8) { atomicAdd(&s[wave][(it * 7 & 15) * 64 + lane], v); }   // 64 lanes, all distinct, consecutive
        v += 1.f;
    }
    __syncthreads();
    float a = 0.f;
    for (int x = tid; x < 4 * 1280; x += 256) a += (&s[0][0])[x];
    out[blockIdx.x * 256 + tid] = a;
}
template <int MODE> void run(const char *name)
{
    float *out; hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2000, blocks = 256 * 5;
    k<MODE><<<blocks, 256>>>(out, iters); hipDeviceSynchronize();
    hipEventRecord(a); k<MODE><<<blocks, 256>>>(out, iters); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // per CU: 5 WGs x 4 waves x iters instructions
    printf("%-28s %.3f ms  -> %.1f clk per wave-instruction per CU (2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / (5.0 * 4 * iters));
    hipFree(out);
}
int main()
{
    run<0>("atomic, 4 distinct splats");
    run<1>("atomic, 4 rows same addr");
    run<2>("atomic, 2-way same addr");
    run<3>("plain store");
    run<4>("plain read-add-write");
    run<5>("atomic f32, 10 lanes");
    run<6>("atomic f32, 1 lane");
    run<7>("atomic u32, 40 lanes");
    run<8>("atomic f32, 64 consecutive");
    return 0;
}
