// Dev microbenchmark (round 6): what does an f32 MFMA issue at when its operands change?  gemm_stream64_f32.hip's rows form runs at 60 % of the
// f32 matrix rate with neither its LDS latency nor its DMA instruction count mattering (profiles/r06_stream64_f32_pipelined.txt).
// Register-resident loops, one wave per SIMD, 64 accumulator blocks (16x16x4) or 16 (32x32x2):
//   ORDER 0: consecutive MFMAs share the FIRST source (srcA), the second changes; 1: share the SECOND (srcB); 2: both change; 3: neither changes
//   (the rows form: pairs share srcA, every pair has a new one)
// build: hipcc --offload-arch=gfx950 -O3 tools/dev/mfma_f32_issue_probe.hip -o tools/dev/mfma_f32_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int ORDER>
__global__ void __launch_bounds__(256) k16(uint32_t iters, float *sink, unsigned long long *clk)
{
    const int tid = threadIdx.x;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = (float)((tid * 977 + i * 131) % 1024) / 512.f - 1.f; b[i] = (float)((tid * 613 + i * 257) % 1024) / 512.f - 1.f; }
    f32x4 acc[8][8];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < 64; ++n) {
            const int i = ORDER == 0 ? n / 8 : ORDER == 1 ? n % 8 : ORDER == 2 ? n % 8 : 0;
            const int j = ORDER == 0 ? n % 8 : ORDER == 1 ? n / 8 : ORDER == 2 ? (n / 8 + n) % 8 : 0;
            const int ai = ORDER == 3 ? n / 8 : i, aj = ORDER == 3 ? n % 8 : j;
            acc[ai][aj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[ai][aj], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    if (s == 1.2345e38f) sink[0] = s;
    if (blockIdx.x == 0 && tid == 0) clk[0] = t1 - t0;
}
template <int ORDER>
__global__ void __launch_bounds__(256) k32(uint32_t iters, float *sink, unsigned long long *clk)
{
    const int tid = threadIdx.x;
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = (float)((tid * 977 + i * 131) % 1024) / 512.f - 1.f; b[i] = (float)((tid * 613 + i * 257) % 1024) / 512.f - 1.f; }
    f32x16 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            const int i = ORDER == 0 ? n / 4 : ORDER == 1 ? n % 4 : n % 4;
            const int j = ORDER == 0 ? n % 4 : ORDER == 1 ? n / 4 : (n / 4 + n) % 4;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 1.2345e38f) sink[0] = s;
    if (blockIdx.x == 0 && tid == 0) clk[0] = t1 - t0;
}
template <typename K> void run(const char *name, K kern, int per_iter, double flop_per)
{
    float *sink; unsigned long long *clk;
    hipMalloc(&sink, 4); hipMalloc(&clk, 16);
    const uint32_t iters = 20000;
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, 200u, sink, clk);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, iters, sink, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    printf("%-34s %6.2f cycles per MFMA   %7.1f TFLOP/s\n", name, (double)h[0] / ((double)iters * per_iter), 256.0 * 4 * iters * per_iter * flop_per / (ms * 1e-3) / 1e12);
}
int main()
{
    run("16x16x4  share srcA", k16<0>, 64, 2048.0);
    run("16x16x4  share srcB", k16<1>, 64, 2048.0);
    run("16x16x4  both change", k16<2>, 64, 2048.0);
    run("16x16x4  neither changes", k16<3>, 64, 2048.0);
    run("32x32x2  share srcA", k32<0>, 16, 4096.0);
    run("32x32x2  share srcB", k32<1>, 16, 4096.0);
    run("32x32x2  both change", k32<2>, 16, 4096.0);
    return 0;
}
