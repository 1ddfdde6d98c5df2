// Dev microbenchmark: per-CU L2->CU streaming rate for (a) global_load_dwordx4 into VGPRs,
// (b) global_load_lds_dwordx4 (LDS-DMA), (c) loads + ds_write_b128.  Every workgroup re-reads a
// small L2-resident window so HBM is out of the picture.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) probe(const char *__restrict__ src, size_t window, int iters, float *sink)
{
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // each workgroup streams its own window (distinct per XCD group of blocks to stay L2 resident)
    const char *base = src + (size_t)(blockIdx.x % 64) * window;
    f32x4 acc = {0, 0, 0, 0};
    const int per_iter = WAVES * 64 * 16 * 8;   // bytes per iteration per workgroup (8 instrs per lane)
    for (int it = 0; it < iters; ++it) {
        const char *p = base + ((size_t)it * per_iter) % window + tid * 16;
        if (MODE == 0) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4 *>(p + u * WAVES * 1024);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(p + u * WAVES * 1024),
                                                 (__attribute__((address_space(3))) void *)(lds + (u * WAVES + wave) * 1024), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4 *>(p + u * WAVES * 1024);
#pragma unroll
            for (int u = 0; u < 8; ++u) *reinterpret_cast<f32x4 *>(lds + (u * WAVES + wave) * 1024 + lane * 16) = v[u];
        }
    }
    if (MODE == 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); acc += *reinterpret_cast<f32x4 *>(lds + tid * 16); }
    if (MODE == 2) { __syncthreads(); acc += *reinterpret_cast<f32x4 *>(lds + tid * 16); }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345e38f) sink[0] = acc[0];
}

template <int MODE, int WAVES>
void run(const char *name, const char *buf, size_t window, float *sink, int grid)
{
    const int iters = 2000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const size_t lds = (size_t)WAVES * 8 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe<MODE, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    probe<MODE, WAVES><<<grid, WAVES * 64, lds>>>(buf, window, 10, sink);
    hipDeviceSynchronize();
    hipEventRecord(a);
    probe<MODE, WAVES><<<grid, WAVES * 64, lds>>>(buf, window, iters, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)grid * iters * WAVES * 64 * 16 * 8;
    printf("%-28s waves/WG %d grid %4d: %.2f TB/s aggregate, %.1f GB/s per CU (%s)\n", name, WAVES, grid, bytes / ms / 1e9,
           bytes / ms / 1e6 / 256, hipGetErrorString(hipGetLastError()));
}

int main()
{
    const size_t window = 256 * 1024;          // per-workgroup window; 64 windows = 16 MiB total (L2/MALL resident)
    char *buf; float *sink;
    hipMalloc(&buf, 64 * window + (1 << 20)); hipMemset(buf, 0, 64 * window + (1 << 20)); hipMalloc(&sink, 64);
    for (int grid : {256, 512}) {
        run<0, 4>("global_load_dwordx4 -> VGPR", buf, window, sink, grid);
        run<0, 8>("global_load_dwordx4 -> VGPR", buf, window, sink, grid);
        run<1, 4>("global_load_lds_dwordx4", buf, window, sink, grid);
        run<1, 8>("global_load_lds_dwordx4", buf, window, sink, grid);
        run<2, 4>("load + ds_write_b128", buf, window, sink, grid);
        run<2, 8>("load + ds_write_b128", buf, window, sink, grid);
    }
    return 0;
}
