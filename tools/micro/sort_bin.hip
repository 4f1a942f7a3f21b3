// Micro-benchmark (GPU box): how long does ONE workgroup of k_sort_tiles take for a bin of n keys?  Includes the library's
// translation unit, so it times the shipped code:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize [-DT4D_SORT_ABL=k]
//   tools/micro/sort_bin.hip -o tools/micro/sort_bin && tools/micro/sort_bin
#include "../../topo4d_amd/csrc/t4d_raster.hip"
#include <algorithm>
#include <random>

int main()
{
    const int sizes[] = { 64, 200, 411, 512, 513, 800, 1286, 2048, 4000, 12614 };
    const uint32_t cap = 1u << 16;
    unsigned long long *d_keys, *d_tmp;
    uint4 *d_items;
    uint32_t *d_fill;
    hipMalloc(&d_keys, cap * 8); hipMalloc(&d_tmp, cap * 8); hipMalloc(&d_items, 16); hipMalloc(&d_fill, kBuckets * 4);
    std::mt19937_64 rng(1);
    for (int wide = 0; wide < 2; wide++)
    for (int n : sizes) {
        if (wide && n > kSortLdsCap) continue;
        std::vector<unsigned long long> h(n);
        for (int i = 0; i < n; i++) h[i] = (rng() & 0x3fffffff00000000ull) | (uint32_t)i;
        KP kp;
        memset(&kp, 0, sizeof(kp));
        kp.V = 1; kp.T = 1; kp.cap = cap; kp.keys = d_keys; kp.sort_tmp = d_tmp; kp.items = d_items; kp.bucket_fill = d_fill;
        kp.long_bins_elsewhere = n > kSortLdsCap ? 1 : 0;
        const uint4 it = make_uint4(0u, 0u, (uint32_t)n, (uint32_t)n);
        uint32_t fill[kBuckets] = { 0 };
        int c = 31 - __builtin_clz(n);
        fill[(kBuckets - 2) - std::min(kBuckets - 2, c)] = 1;
        hipMemcpy(d_items, &it, 16, hipMemcpyHostToDevice);
        hipMemcpy(d_fill, fill, sizeof(fill), hipMemcpyHostToDevice);
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        float best = 1e9f;
        for (int rep = 0; rep < 5; rep++) {
            hipMemcpy(d_keys, h.data(), n * 8, hipMemcpyHostToDevice);
            for (int w = 0; w < 20; w++) hipLaunchKernelGGL(k_sort_tiles<kBlock>, dim3(1), dim3(kBlock), 0, 0, kp);     // warm clocks
            hipEventRecord(a, 0);
            const int iters = 50;
            for (int i = 0; i < iters; i++) {
                if (n > kSortLdsCap) hipLaunchKernelGGL(k_sort_long, dim3(1), dim3(kLongBlock), 0, 0, kp);
                else if (wide) hipLaunchKernelGGL(k_sort_tiles<kLongBlock>, dim3(1), dim3(kLongBlock), 0, 0, kp);
                else hipLaunchKernelGGL(k_sort_tiles<kBlock>, dim3(1), dim3(kBlock), 0, 0, kp);
            }
            hipEventRecord(b, 0);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            best = std::min(best, ms * 1000.f / iters);
        }
        std::vector<unsigned long long> out(n);
        hipMemcpy(out.data(), d_keys, n * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        printf("%4d threads, n = %5d: %7.2f us per launch (back-to-back launches of one workgroup), %s\n", wide ? kLongBlock : (n > kSortLdsCap ? kLongBlock : kBlock), n, best, out == h ? "sorted" : "WRONG ORDER");
    }
    return 0;
}
