// Dev microbenchmark (round 5): does the narrow-shape / stationary-srcB result of tools/dev/mfma_issue_probe.hip carry over to fp8?
// Register-resident loops of v_mfma_f32_32x32x64_f8f6f4 (4 x 4 blocks, what gemm_lp256w4.hip issues) and v_mfma_f32_16x16x128_f8f6f4
// (8 x 8 blocks) on e4m3 operands, all-ones and uniform[-1,1), in the two issue orders.
// build: hipcc --offload-arch=gfx950 -O3 mfma_fp8_shape_probe.hip -o /tmp/fp8p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
__device__ inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
// e4m3 bytes: ones = 0x38; "uniform" = random sign, exponent 0..7 biased low, random mantissa: |x| < 1, no NaN (0x7F / 0xFF excluded by exponent <= 7)
__device__ inline i32x8 rnd(uint32_t seed, bool ones)
{
    i32x8 v;
    for (int e = 0; e < 8; ++e) {
        uint32_t w = 0;
        for (int b = 0; b < 4; ++b) {
            const uint32_t r = mix(seed * 32 + e * 4 + b);
            // uniform[-1,1) rounded to e4m3: pick the value by converting a float
            const float f = (r >> 8) * (2.0f / 16777216.0f) - 1.0f;
            uint32_t byte;
            if (ones) byte = 0x38;
            else {
                const float a = fabsf(f);
                int ex; const float m = frexpf(a, &ex);                 // a = m 2^ex, m in [0.5, 1)
                int E = ex - 1 + 7;                                      // biased exponent of 1.xxx form
                uint32_t mant;
                if (a < 0.001953125f) { E = 0; mant = (uint32_t)(a * 512.0f + 0.5f); if (mant > 7) { E = 1; mant = 0; } }
                else if (E <= 0) { mant = (uint32_t)(a * 512.0f + 0.5f); E = 0; if (mant > 7) { E = 1; mant = 0; } }
                else { mant = (uint32_t)((m * 2.0f - 1.0f) * 8.0f + 0.5f); if (mant > 7) { mant = 0; ++E; } }
                byte = (f < 0 ? 0x80u : 0u) | ((uint32_t)E << 3) | mant;
            }
            w |= byte << (8 * b);
        }
        v[e] = (int)w;
    }
    return v;
}
template <int ORDER, bool ONES>
__global__ void __launch_bounds__(256) k16(uint32_t iters, float *sink, unsigned long long *clk)
{
    const int tid = threadIdx.x;
    constexpr int NI = 8, NJ = 8;
    i32x8 a[NI], b[NJ];
    for (int i = 0; i < NI; ++i) a[i] = rnd(tid * 977 + i * 131 + blockIdx.x * 7919, ONES);
    for (int j = 0; j < NJ; ++j) b[j] = rnd(tid * 613 + j * 257 + 99991 + blockIdx.x * 104729, ONES);
    f32x4 acc[NI][NJ];
    for (int i = 0; i < NI; ++i) for (int j = 0; j < NJ; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NI * NJ; ++n) {
            const int i = ORDER == 0 ? n % NI : n / NJ, j = ORDER == 0 ? n / NI : n % NJ;
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b[j], a[i], acc[i][j], 0, 0, 0, 0, 0, 0);
        }
        i32x8 t = a[0];
#pragma unroll
        for (int i = 0; i + 1 < NI; ++i) a[i] = a[i + 1];
        a[NI - 1] = t;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < NI; ++i) for (int j = 0; j < NJ; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    if (s == 1.2345e38f) sink[0] = s;
    if (blockIdx.x == 0 && tid == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
template <int ORDER, bool ONES>
__global__ void __launch_bounds__(256) k32(uint32_t iters, float *sink, unsigned long long *clk)
{
    const int tid = threadIdx.x;
    constexpr int NI = 4, NJ = 4;
    i32x8 a[NI], b[NJ];
    for (int i = 0; i < NI; ++i) a[i] = rnd(tid * 977 + i * 131 + blockIdx.x * 7919, ONES);
    for (int j = 0; j < NJ; ++j) b[j] = rnd(tid * 613 + j * 257 + 99991 + blockIdx.x * 104729, ONES);
    f32x16 acc[NI][NJ];
    for (int i = 0; i < NI; ++i) for (int j = 0; j < NJ; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NI * NJ; ++n) {
            const int i = ORDER == 0 ? n % NI : n / NJ, j = ORDER == 0 ? n / NI : n % NJ;
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b[j], a[i], acc[i][j], 0, 0, 0, 0, 0, 0);
        }
        i32x8 t = a[0];
#pragma unroll
        for (int i = 0; i + 1 < NI; ++i) a[i] = a[i + 1];
        a[NI - 1] = t;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < NI; ++i) for (int j = 0; j < NJ; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 1.2345e38f) sink[0] = s;
    if (blockIdx.x == 0 && tid == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
template <typename K> void run(const char *name, K kern, double flop_per_iter_per_wave, int mfma_per_iter)
{
    float *sink; unsigned long long *clk, h[2];
    hipMalloc(&sink, 4); hipMalloc(&clk, 16);
    const uint32_t iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, 2000, sink, clk);          // warm-up: reach the sustained clock
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, iters, sink, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double tf = flop_per_iter_per_wave * 1024.0 * iters / (ms * 1e-3) / 1e12;
    const double ghz = (double)h[0] / ((double)h[1] / 100e6) / 1e9;
    printf("%-58s %6.0f TF  %.3f GHz  %6.2f shader cycles per MFMA and SIMD\n", name, tf, ghz, (double)h[0] / ((double)iters * mfma_per_iter));
    hipFree(sink); hipFree(clk);
}
int main()
{
    const double f16 = 64.0 * 2 * 16 * 16 * 128, f32 = 16.0 * 2 * 32 * 32 * 64;
    printf("== all ones (e4m3)\n");
    run("32x32x64  order j/i (consecutive MFMAs share srcA)", k32<0, true>, f32, 16);
    run("32x32x64  order i/j (share srcB)", k32<1, true>, f32, 16);
    run("16x16x128 order j/i (share srcA)", k16<0, true>, f16, 64);
    run("16x16x128 order i/j (share srcB)", k16<1, true>, f16, 64);
    printf("== uniform[-1,1) rounded to e4m3\n");
    run("32x32x64  order j/i (consecutive MFMAs share srcA)", k32<0, false>, f32, 16);
    run("32x32x64  order i/j (share srcB)", k32<1, false>, f32, 16);
    run("16x16x128 order j/i (share srcA)", k16<0, false>, f16, 64);
    run("16x16x128 order i/j (share srcB)", k16<1, false>, f16, 64);
    return 0;
}
