// Dev microbenchmark (round 5, review item "one more C3 lever"): v_mfma_f32_16x16x32_bf16 holds 2.2 GHz on uniform operands where the
// 32x32x16 shape falls to 1.88, but issued at 18.9 cycles where 16 are nominal (profiles/r02_c3_data_movement_levers.md (e)).  Where do
// the 2.9 cycles go?  Variants of the register-resident loop of tools/dev/mfma_shape_probe.hip:
//   ORDER 0  j outer, i inner (consecutive MFMAs share the B fragment; an accumulator comes round every 64 MFMAs) -- the r02 probe
//   ORDER 1  i outer, j inner (share the A fragment)
//   ORDER 2  diagonal: MFMA n uses (i, j) = (n % 8, (n / 8 + n) % 8): both fragments change every instruction
//   ROT 0    operands never change (no v_mov between k-steps); ROT 1 the A fragments rotate every k-step (the r02 probe)
//   WPS 2    two waves per SIMD, each an 8 x 4 block of accumulators (128 registers), same FLOPs per SIMD
// build: hipcc --offload-arch=gfx950 -O3 mfma_issue_probe.hip -o /tmp/mip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ inline bf16x8 rnd(uint32_t seed, bool ones)
{
    bf16x8 v;
    for (int e = 0; e < 8; ++e) v[e] = (__bf16)(ones ? 1.f : (mix(seed * 8 + e) >> 8) * (2.0f / 16777216.0f) - 1.0f);
    return v;
}
template <int ORDER, int ROT, int WPS, bool ONES>
__global__ void __launch_bounds__(256 * WPS) k16(uint32_t iters, float *sink, unsigned long long *clk)
{
    const int tid = threadIdx.x;
    constexpr int NI = 8, NJ = 8 / WPS;
    bf16x8 a[NI], b[NJ];
    for (int i = 0; i < NI; ++i) a[i] = rnd(tid * 977 + i * 131 + blockIdx.x * 7919, ONES);
    for (int j = 0; j < NJ; ++j) b[j] = rnd(tid * 613 + j * 257 + 99991 + blockIdx.x * 104729, ONES);
    f32x4 acc[NI][NJ];
    for (int i = 0; i < NI; ++i) for (int j = 0; j < NJ; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NI * NJ; ++n) {
            const int i = ORDER == 0 ? n % NI : ORDER == 1 ? n / NJ : n % NI;
            const int j = ORDER == 0 ? n / NI : ORDER == 1 ? n % NJ : (n / NI + n) % NJ;
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
        }
        if (ROT) {
            bf16x8 t = a[0];
#pragma unroll
            for (int i = 0; i + 1 < NI; ++i) a[i] = a[i + 1];
            a[NI - 1] = t;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < NI; ++i) for (int j = 0; j < NJ; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    if (s == 1.2345e38f) sink[0] = s;
    if (blockIdx.x == 0 && tid == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
// the 32x32x16 shape with the same switches (4 x 4 blocks; WPS 2: 4 x 2 per wave)
template <int ORDER, int ROT, int WPS, bool ONES>
__global__ void __launch_bounds__(256 * WPS) k32(uint32_t iters, float *sink, unsigned long long *clk)
{
    const int tid = threadIdx.x;
    constexpr int NI = 4, NJ = 4 / WPS;
    bf16x8 a[NI], b[NJ];
    for (int i = 0; i < NI; ++i) a[i] = rnd(tid * 977 + i * 131 + blockIdx.x * 7919, ONES);
    for (int j = 0; j < NJ; ++j) b[j] = rnd(tid * 613 + j * 257 + 99991 + blockIdx.x * 104729, ONES);
    f32x16 acc[NI][NJ];
    for (int i = 0; i < NI; ++i) for (int j = 0; j < NJ; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NI * NJ; ++n) {
            const int i = ORDER == 0 ? n % NI : ORDER == 1 ? n / NJ : n % NI;
            const int j = ORDER == 0 ? n / NI : ORDER == 1 ? n % NJ : (n / NI + n) % NJ;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
        }
        if (ROT) {
            bf16x8 t = a[0];
#pragma unroll
            for (int i = 0; i + 1 < NI; ++i) a[i] = a[i + 1];
            a[NI - 1] = t;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < NI; ++i) for (int j = 0; j < NJ; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 1.2345e38f) sink[0] = s;
    if (blockIdx.x == 0 && tid == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
template <typename K> void run(const char *name, K kern, int wps, int shape, float *sink, unsigned long long *clk)
{
    const int nmf = shape == 16 ? 64 / wps : 16 / wps;                   // MFMAs per wave and iteration
    const uint32_t iters = shape == 16 ? 20000 : 40000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    kern<<<256, 256 * wps>>>(2000, sink, clk); (void)hipDeviceSynchronize();
    double best_tf = 0, ghz = 0, cyc = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        kern<<<256, 256 * wps>>>(iters, sink, clk);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        const double flop = 256.0 * 4 * wps * iters * nmf * (shape == 16 ? 16384.0 : 32768.0);
        const double tf = flop / ms / 1e9;
        if (tf > best_tf) { best_tf = tf; ghz = (double)h[0] / ((double)h[1] / 100e6) / 1e9; cyc = (double)h[0] / ((double)iters * nmf * wps); }
    }
    printf("%-58s %7.0f TF  %.3f GHz  %5.2f shader cycles per MFMA and SIMD\n", name, best_tf, ghz, cyc);
}
#define RUN16(O, R, W, ONES, NAME) run(NAME, k16<O, R, W, ONES>, W, 16, sink, clk)
#define RUN32(O, R, W, ONES, NAME) run(NAME, k32<O, R, W, ONES>, W, 32, sink, clk)
int main()
{
    float *sink; unsigned long long *clk;
    (void)hipMalloc(&sink, 64); (void)hipMalloc(&clk, 64);
    for (int ones = 1; ones >= 0; --ones) {
        printf("== %s\n", ones ? "all ones" : "uniform[-1,1)");
        if (ones) {
            RUN32(0, 1, 1, true, "32x32x16 order j/i, rotating A (the r02 probe)"); RUN16(0, 1, 1, true, "16x16x32 order j/i, rotating A (the r02 probe)");
            RUN16(0, 0, 1, true, "16x16x32 order j/i, fixed operands"); RUN16(1, 1, 1, true, "16x16x32 order i/j, rotating A");
            RUN16(2, 1, 1, true, "16x16x32 diagonal order, rotating A"); RUN16(2, 0, 1, true, "16x16x32 diagonal order, fixed operands");
            RUN16(0, 1, 2, true, "16x16x32 order j/i, rotating A, two waves per SIMD"); RUN16(2, 1, 2, true, "16x16x32 diagonal, rotating A, two waves per SIMD");
            RUN32(0, 1, 2, true, "32x32x16 order j/i, rotating A, two waves per SIMD");
        } else {
            RUN32(0, 1, 1, false, "32x32x16 order j/i, rotating A (the r02 probe)"); RUN16(0, 1, 1, false, "16x16x32 order j/i, rotating A (the r02 probe)");
            RUN16(0, 0, 1, false, "16x16x32 order j/i, fixed operands"); RUN16(1, 1, 1, false, "16x16x32 order i/j, rotating A");
            RUN16(2, 1, 1, false, "16x16x32 diagonal order, rotating A"); RUN16(2, 0, 1, false, "16x16x32 diagonal order, fixed operands");
            RUN16(0, 1, 2, false, "16x16x32 order j/i, rotating A, two waves per SIMD"); RUN16(2, 1, 2, false, "16x16x32 diagonal, rotating A, two waves per SIMD");
            RUN32(0, 1, 2, false, "32x32x16 order j/i, rotating A, two waves per SIMD");
            // (late in round 5, after the fp8 probe showed the order mattering for the wide shape too)
            RUN32(1, 1, 1, false, "32x32x16 order i/j (share srcB), rotating A"); RUN32(1, 0, 1, false, "32x32x16 order i/j, fixed operands");
            RUN32(0, 0, 1, false, "32x32x16 order j/i, fixed operands"); RUN32(2, 1, 1, false, "32x32x16 diagonal order, rotating A");
            RUN32(0, 1, 1, false, "32x32x16 order j/i, rotating A (again)"); RUN32(1, 1, 1, false, "32x32x16 order i/j, rotating A (again)");
        }
    }
    return 0;
}
