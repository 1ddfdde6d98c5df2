// Dev probe: how fast does HBM deliver a COLD row-major [8192 k][16 KiB] matrix when every workgroup owns a COLUMN STRIP
// (what a few-rows product x [M][K] * W [K][N] with a row-major W asks for): strips of 64 / 128 / 256 bytes, K cut so that
// 256 or 512 workgroups exist, with / without keeping neighbouring strips on one XCD, with / without a K stagger.
// Eight 128 MiB matrices are rotated (cold operands).  GB/s, median of 24 launches.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int KROWS = 8192, ROW_BYTES = 16384;

// STRIP bytes per k-row and workgroup; KSPLIT slices of K; XCD: neighbouring strips on one XCD; NT loads
template <int STRIP, int KSPLIT, bool XCD, bool NT, bool STAGGER, int UNROLL>
__global__ void __launch_bounds__(256) rd(const char *__restrict__ mat, float *sink)
{
    constexpr int NSTRIP = ROW_BYTES / STRIP, LPR = STRIP / 16, RPI = 64 / LPR;     // lanes per k-row, k-rows per wave instruction
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int b = blockIdx.x;
    if (XCD) { const int per = gridDim.x / 8; b = (b % 8) * per + b / 8; }          // consecutive logical ids share an XCD
    const int strip = b % NSTRIP, slice = b / NSTRIP;
    constexpr int KPER = KROWS / KSPLIT, STEPS = KPER / (4 * RPI);                   // steps per wave
    const char *base = mat + (size_t)slice * KPER * ROW_BYTES + (size_t)strip * STRIP + (lane % LPR) * 16;
    const int shift = STAGGER ? (strip * 7) % STEPS : 0;
    f32x4 acc = {0, 0, 0, 0};
    for (int s0 = 0; s0 < STEPS; s0 += UNROLL) {
        f32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            int s = s0 + u + shift;
            if (s >= STEPS) s -= STEPS;
            const char *p = base + (size_t)((s * 4 + w) * RPI + lane / LPR) * ROW_BYTES;
            v[u] = NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p)) : *reinterpret_cast<const f32x4 *>(p);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345e38f) sink[0] = acc[0];
}

template <int STRIP, int KSPLIT, bool XCD, bool NT, bool STAGGER, int UNROLL>
void run(char **mats, int nm, float *sink)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    std::vector<float> ms;
    const int grid = ROW_BYTES / STRIP * KSPLIT;
    for (int i = 0; i < 28; ++i) {
        hipEventRecord(a);
        rd<STRIP, KSPLIT, XCD, NT, STAGGER, UNROLL><<<grid, 256>>>(mats[i % nm], sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float t; hipEventElapsedTime(&t, a, b);
        if (i >= 4) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    const double bytes = (double)KROWS * ROW_BYTES;
    printf("strip %3d B x K/%d (%4d workgroups) %s %s %s unroll %2d: median %7.1f GB/s (%6.1f us)  best %7.1f\n", STRIP, KSPLIT, grid,
           XCD ? "xcd-paired" : "round-robin", NT ? "nt   " : "plain", STAGGER ? "staggered" : "aligned  ", UNROLL,
           bytes / ms[ms.size() / 2] / 1e6, ms[ms.size() / 2] * 1e3, bytes / ms[0] / 1e6);
}

int main()
{
    constexpr int NM = 8;
    char *mats[NM]; float *sink; hipMalloc(&sink, 64);
    for (int i = 0; i < NM; ++i) { hipMalloc(&mats[i], (size_t)KROWS * ROW_BYTES); hipMemset(mats[i], 0, (size_t)KROWS * ROW_BYTES); }
    run<64, 1, false, false, false, 16>(mats, NM, sink);  run<64, 1, true, false, false, 16>(mats, NM, sink);
    run<64, 1, false, false, true, 16>(mats, NM, sink);   run<64, 1, true, false, true, 16>(mats, NM, sink);
    run<64, 2, true, false, true, 16>(mats, NM, sink);
    run<128, 2, false, false, false, 16>(mats, NM, sink); run<128, 2, false, false, true, 16>(mats, NM, sink);
    run<128, 2, true, false, true, 16>(mats, NM, sink);   run<128, 4, false, false, true, 16>(mats, NM, sink);
    run<128, 4, true, false, true, 16>(mats, NM, sink);   run<128, 2, false, true, true, 16>(mats, NM, sink);
    run<256, 4, false, false, false, 16>(mats, NM, sink); run<256, 4, false, false, true, 16>(mats, NM, sink);
    run<256, 8, false, false, true, 16>(mats, NM, sink);  run<256, 4, false, true, true, 16>(mats, NM, sink);
    run<512, 8, false, false, true, 16>(mats, NM, sink);  run<512, 16, false, false, true, 16>(mats, NM, sink);
    run<128, 2, false, false, true, 32>(mats, NM, sink);  run<256, 4, false, false, true, 32>(mats, NM, sink);
    return 0;
}
