// Dev probe: how fast does one L2 serve returning agent-scope fetch_adds on ONE address (the arrival ticket / a work queue
// head), from 1 lane per wave, on an otherwise idle chip?  build: hipcc --offload-arch=gfx950 -O3 atomic_rate_probe.hip -o /tmp/arp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(256) k(unsigned int *ctr, int iters, int nctr, unsigned long long *sink)
{
    typedef __attribute__((address_space(1))) unsigned int gu32;
    unsigned int acc = 0;
    if ((threadIdx.x & 63) == 0) {
        unsigned int *p = ctr + (size_t)(blockIdx.x % nctr) * 64;      // counters 256 bytes apart
        for (int i = 0; i < iters; ++i) acc += __hip_atomic_fetch_add((gu32 *)p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (acc == 0xFFFFFFFFu) sink[0] = acc;
    }
}
int main()
{
    unsigned int *ctr; unsigned long long *sink;
    hipMalloc(&ctr, 1 << 20); hipMalloc(&sink, 64); hipMemset(ctr, 0, 1 << 20);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int nctr : {1, 8, 64})
        for (int grid : {8, 64, 256, 768})
            for (int iters : {1, 16}) {
                for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, ctr, iters, nctr, sink);
                hipDeviceSynchronize();
                hipEventRecord(a);
                for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, ctr, iters, nctr, sink);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                const double us = ms * 100.0, total = (double)grid * 4 * iters;
                printf("counters %2d grid %4d x 4 waves x %2d adds: %8.2f us per launch, %7.1f ns per add (%.0f adds)\n", nctr, grid, iters, us, us * 1e3 / total, total);
            }
    return 0;
}
