// Dev microbenchmark: the matrix pipe at the chip's power limit for the two bf16 MFMA shapes, register-resident uniform[-1,1)
// operands, one wave per SIMD computing a 128x128 block as the GEMM does: 4x4 blocks of v_mfma_f32_32x32x16_bf16 (16 MFMAs per
// k-step of 16) against 8x8 blocks of v_mfma_f32_16x16x32_bf16 (64 MFMAs per k-step of 32).  Same FLOPs, same operand bytes
// from the register file per k; the 16x16 shape reads and writes half as many accumulator bytes per FLOP.  Question: does the
// sustained clock (and so the GEMM's ceiling on this operand distribution) differ?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ inline bf16x8 rnd(uint32_t seed, bool ones)
{
    bf16x8 v;
    for (int e = 0; e < 8; ++e) v[e] = (__bf16)(ones ? 1.f : (mix(seed * 8 + e) >> 8) * (2.0f / 16777216.0f) - 1.0f);
    return v;
}

template <int SHAPE, bool ONES>
__global__ void __launch_bounds__(256) k(uint32_t iters, float *sink, unsigned long long *clk)
{
    const int tid = threadIdx.x;
    constexpr int NF = SHAPE == 32 ? 4 : 8;
    bf16x8 a[NF], b[NF];
    for (int i = 0; i < NF; ++i) { a[i] = rnd(tid * 977 + i * 131 + blockIdx.x * 7919, ONES); b[i] = rnd(tid * 613 + i * 257 + 99991 + blockIdx.x * 104729, ONES); }
    f32x16 acc32[SHAPE == 32 ? 4 : 1][SHAPE == 32 ? 4 : 1];
    f32x4 acc16[SHAPE == 16 ? 8 : 1][SHAPE == 16 ? 8 : 1];
    if constexpr (SHAPE == 32) { for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc32[i][j][r] = 0.f; }
    else { for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) for (int r = 0; r < 4; ++r) acc16[i][j][r] = 0.f; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (uint32_t it = 0; it < iters; ++it) {
        if constexpr (SHAPE == 32) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc32[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc32[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc16[i][j], 0, 0, 0);
        }
        bf16x8 t = a[0];
#pragma unroll
        for (int i = 0; i + 1 < NF; ++i) a[i] = a[i + 1];
        a[NF - 1] = t;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    if constexpr (SHAPE == 32) { for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc32[i][j][r]; }
    else { for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) for (int r = 0; r < 4; ++r) s += acc16[i][j][r]; }
    if (s == 1.2345e38f) sink[0] = s;
    if (blockIdx.x == 0 && tid == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int SHAPE, bool ONES> void run(const char *name, float *sink, unsigned long long *clk)
{
    // per iteration and wave: 32x32x16 block 4x4 -> k = 16; 16x16x32 block 8x8 -> k = 32: twice the FLOPs per iteration
    const uint32_t iters = SHAPE == 32 ? 40000 : 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<SHAPE, ONES><<<256, 256>>>(2000, sink, clk); hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        k<SHAPE, ONES><<<256, 256>>>(iters, sink, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        const double flop = 256.0 * 4 * iters * 2.0 * 128 * 128 * (SHAPE == 32 ? 16 : 32);
        printf("%-44s %.3f ms  %.0f TF  shader clock %.3f GHz\n", name, ms, flop / ms / 1e9, (double)h[0] / ((double)h[1] / 100e6) / 1e9);
    }
}

int main()
{
    float *sink; unsigned long long *clk;
    hipMalloc(&sink, 64); hipMalloc(&clk, 64);
    for (int round = 0; round < 2; ++round) {
        run<32, true>("32x32x16, all ones", sink, clk);
        run<16, true>("16x16x32, all ones", sink, clk);
        run<32, false>("32x32x16, uniform[-1,1)", sink, clk);
        run<16, false>("16x16x32, uniform[-1,1)", sink, clk);
    }
    return 0;
}
