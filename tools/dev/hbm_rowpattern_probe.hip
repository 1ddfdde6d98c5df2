// Dev probe: how fast does HBM deliver a COLD [rows][16 KiB] matrix when every workgroup owns 32 rows and walks K --
// (a) 128 bytes of each of 8 rows per wave instruction (gemm_stream64's LDS-DMA piece), (b) 1 KiB of ONE row per wave
// instruction, each with plain and non-temporal loads and with / without the per-workgroup K stagger.  Eight 128 MiB
// matrices are rotated so that no launch finds its operand in the 256 MiB Infinity Cache.  GB/s, median of 24 launches.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int ROWS = 8192, ROW_BYTES = 16384, WG_ROWS = 32;

template <int PATTERN, bool NT, bool STAGGER, int UNROLL>
__global__ void __launch_bounds__(256) rd(const char *__restrict__ mat, float *sink)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const char *base = mat + (size_t)blockIdx.x * WG_ROWS * ROW_BYTES;
    f32x4 acc = {0, 0, 0, 0};
    constexpr int STEPS = PATTERN == 0 ? ROW_BYTES / 128 : (ROW_BYTES / 1024) * 8;     // per wave
    const int shift = STAGGER ? (PATTERN == 0 ? blockIdx.x % STEPS : (blockIdx.x % (ROW_BYTES / 1024)) * 8) : 0;
    for (int s0 = 0; s0 < STEPS; s0 += UNROLL) {
        f32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            int s = s0 + u + shift;
            if (s >= STEPS) s -= STEPS;
            const char *p;
            if (PATTERN == 0) p = base + (size_t)(w * 8 + (lane >> 3)) * ROW_BYTES + s * 128 + (lane & 7) * 16;       // 8 rows x 128 B
            else p = base + (size_t)(w * 8 + (s & 7)) * ROW_BYTES + (s >> 3) * 1024 + lane * 16;                       // 1 row x 1 KiB
            v[u] = NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p)) : *reinterpret_cast<const f32x4 *>(p);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345e38f) sink[0] = acc[0];
}

template <int PATTERN, bool NT, bool STAGGER, int UNROLL>
void run(char **mats, int nm, float *sink)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    std::vector<float> ms;
    for (int i = 0; i < 28; ++i) {
        hipEventRecord(a);
        rd<PATTERN, NT, STAGGER, UNROLL><<<ROWS / WG_ROWS, 256>>>(mats[i % nm], sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float t; hipEventElapsedTime(&t, a, b);
        if (i >= 4) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    const double bytes = (double)ROWS * ROW_BYTES;
    printf("%s %s %s unroll %2d: median %7.1f GB/s (%6.1f us)  best %7.1f GB/s\n", PATTERN == 0 ? "8 rows x 128 B" : "1 row x 1 KiB  ",
           NT ? "nt   " : "plain", STAGGER ? "staggered" : "aligned  ", UNROLL, bytes / ms[ms.size() / 2] / 1e6, ms[ms.size() / 2] * 1e3, bytes / ms[0] / 1e6);
}

int main()
{
    constexpr int NM = 8;
    char *mats[NM]; float *sink; hipMalloc(&sink, 64);
    for (int i = 0; i < NM; ++i) { hipMalloc(&mats[i], (size_t)ROWS * ROW_BYTES); hipMemset(mats[i], 0, (size_t)ROWS * ROW_BYTES); }
    run<0, false, true, 16>(mats, NM, sink);  run<0, true, true, 16>(mats, NM, sink);
    run<1, false, true, 16>(mats, NM, sink);  run<1, true, true, 16>(mats, NM, sink);
    run<0, false, false, 16>(mats, NM, sink); run<1, false, false, 16>(mats, NM, sink);
    run<0, false, true, 32>(mats, NM, sink);  run<1, false, true, 32>(mats, NM, sink);
    run<0, true, true, 32>(mats, NM, sink);   run<1, true, true, 32>(mats, NM, sink);
    printf("-- warm (one matrix)\n");
    run<0, false, true, 16>(mats, 1, sink);   run<1, false, true, 16>(mats, 1, sink);
    return 0;
}
