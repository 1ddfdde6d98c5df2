// Dev microbenchmark: MFMA 32x32x16 bf16 issue rate with (a) all-ones operands, (b) random uniform[-1,1)
// operands, register-resident (no LDS / memory traffic): isolates how much of the GEMM's clock loss is
// the matrix pipe's own data-dependent power.  MODE 2 adds the GEMM's LDS fragment-read traffic.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE>
__global__ void __launch_bounds__(256) k(uint32_t iters, float *sink, unsigned long long *clk)
{
    __shared__ __attribute__((aligned(16))) char lds[64 * 1024];
    const int tid = threadIdx.x;
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 8; ++e) {
            float va = 1.f, vb = 1.f;
            if (MODE >= 1) {
                va = (mix(tid * 977 + i * 131 + e * 7 + blockIdx.x * 7919) >> 8) * (2.0f / 16777216.0f) - 1.0f;
                vb = (mix(tid * 613 + i * 257 + e * 11 + 99991 + blockIdx.x * 104729) >> 8) * (2.0f / 16777216.0f) - 1.0f;
            }
            a[i][e] = (__bf16)va; b[i][e] = (__bf16)vb;
        }
    if (MODE == 2) {
        for (int i = tid; i < 64 * 1024 / 16; i += 256) {
            bf16x8 v;
            for (int e = 0; e < 8; ++e) v[e] = (__bf16)((mix(i * 8 + e) >> 8) * (2.0f / 16777216.0f) - 1.0f);
            *reinterpret_cast<bf16x8 *>(lds + i * 16) = v;
        }
        __syncthreads();
    }
    f32x16 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (uint32_t it = 0; it < iters; ++it) {
        if (MODE == 2) {
            const int off = ((it * 8192) & 0xFFFF) + (tid & 63) * 16;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = *reinterpret_cast<bf16x8 *>(lds + ((off + i * 1024) & 0xFFFF));
                b[i] = *reinterpret_cast<bf16x8 *>(lds + ((off + 4096 + i * 1024) & 0xFFFF));
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
        if (MODE == 1) {   // rotate operands so consecutive MFMAs see different data, as in the GEMM
            bf16x8 t = a[0]; a[0] = a[1]; a[1] = a[2]; a[2] = a[3]; a[3] = t;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float t = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) t += acc[i][j][r];
    if (t == 1.2345e38f) sink[0] = t;
    if (blockIdx.x == 0 && tid == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int MODE> void run(const char *name, float *sink, unsigned long long *clk)
{
    const uint32_t iters = 40000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256, 256>>>(2000, sink, clk); hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        k<MODE><<<256, 256>>>(iters, sink, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        const double flop = 256.0 * 4 * iters * 16 * 2.0 * 32 * 32 * 16;
        printf("%-34s %.3f ms  %.0f TF  shader clock %.3f GHz (memtime/realtime@100MHz)  cycles/MFMA %.2f\n", name, ms, flop / ms / 1e9,
               (double)h[0] / ((double)h[1] / 100e6) / 1e9, (double)h[0] / (iters * 16.0));
    }
}

int main()
{
    float *sink; unsigned long long *clk; hipMalloc(&sink, 64); hipMalloc(&clk, 64);
    run<0>("ones, registers", sink, clk);
    run<1>("uniform[-1,1), registers", sink, clk);
    run<2>("uniform[-1,1), + LDS fragment reads", sink, clk);
    run<0>("ones, registers (again)", sink, clk);
    return 0;
}
