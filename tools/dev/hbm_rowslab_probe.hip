// Dev probe (round 4): the wide end of hbm_colstrip_probe.hip.  A COLD row-major [8192 k][16 KiB] matrix read by workgroups
// that own STRIP bytes of every k-row of a K slice, STRIP = 1 KiB ... the whole 16 KiB row (a row slab: fully contiguous),
// K cut so that 256 / 512 / 1024 workgroups exist.  What a few-rows product against a row-major weight could stream at if
// it paid for the wider partial vectors that wide strips imply.  Eight 128 MiB matrices are rotated (cold operands).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int KROWS = 8192, ROW_BYTES = 16384;

template <int STRIP, int KSPLIT, bool NT, int UNROLL, bool SLICE_MAJOR>
__global__ void __launch_bounds__(256) rd(const char *__restrict__ mat, float *sink)
{
    constexpr int NSTRIP = ROW_BYTES / STRIP, KPER = KROWS / KSPLIT;
    constexpr int PASS = 4096;                                            // bytes one workgroup instruction round covers
    constexpr int STEPS = KPER * STRIP / PASS;
    const int t = threadIdx.x;
    // SLICE_MAJOR: consecutive workgroups walk the strips of one K slice (neighbours read neighbouring columns of the same
    // rows); else consecutive workgroups take consecutive K slices of one strip
    const int strip = SLICE_MAJOR ? blockIdx.x % NSTRIP : blockIdx.x / KSPLIT;
    const int slice = SLICE_MAJOR ? blockIdx.x / NSTRIP : blockIdx.x % KSPLIT;
    const char *base = mat + (size_t)slice * KPER * ROW_BYTES + (size_t)strip * STRIP;
    f32x4 acc = {0, 0, 0, 0};
    for (int s0 = 0; s0 < STEPS; s0 += UNROLL) {
        f32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int s = s0 + u;
            size_t off;
            if (STRIP >= PASS) off = (size_t)(s / (STRIP / PASS)) * ROW_BYTES + (size_t)(s % (STRIP / PASS)) * PASS + t * 16;
            else off = (size_t)(s * (PASS / STRIP) + t * 16 / STRIP) * ROW_BYTES + (t * 16) % STRIP;
            const char *p = base + off;
            v[u] = NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p)) : *reinterpret_cast<const f32x4 *>(p);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345e38f) sink[0] = acc[0];
}

template <int STRIP, int KSPLIT, bool NT, int UNROLL, bool SLICE_MAJOR>
void run(char **mats, int nm, float *sink)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    std::vector<float> ms;
    const int grid = ROW_BYTES / STRIP * KSPLIT;
    for (int i = 0; i < 28; ++i) {
        hipEventRecord(a);
        rd<STRIP, KSPLIT, NT, UNROLL, SLICE_MAJOR><<<grid, 256>>>(mats[i % nm], sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float t; hipEventElapsedTime(&t, a, b);
        if (i >= 4) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    const double bytes = (double)KROWS * ROW_BYTES;
    printf("strip %5d B x K/%-4d (%4d workgroups, %4d rows each) %s %s unroll %2d: median %7.1f GB/s (%6.1f us)  best %7.1f\n", STRIP, KSPLIT,
           grid, KROWS / KSPLIT, SLICE_MAJOR ? "slice-major" : "strip-major", NT ? "nt   " : "plain", UNROLL,
           bytes / ms[ms.size() / 2] / 1e6, ms[ms.size() / 2] * 1e3, bytes / ms[0] / 1e6);
}

template <int STRIP, int KSPLIT>
void both(char **mats, int nm, float *sink)
{
    run<STRIP, KSPLIT, true, 8, true>(mats, nm, sink);
    run<STRIP, KSPLIT, true, 8, false>(mats, nm, sink);
    run<STRIP, KSPLIT, false, 8, true>(mats, nm, sink);
}

int main()
{
    constexpr int NM = 8;
    char *mats[NM]; float *sink; hipMalloc(&sink, 64);
    for (int i = 0; i < NM; ++i) { hipMalloc(&mats[i], (size_t)KROWS * ROW_BYTES); hipMemset(mats[i], 0, (size_t)KROWS * ROW_BYTES); }
    both<1024, 16>(mats, NM, sink);  both<1024, 32>(mats, NM, sink);  both<1024, 64>(mats, NM, sink);
    both<2048, 32>(mats, NM, sink);  both<2048, 64>(mats, NM, sink);  both<2048, 128>(mats, NM, sink);
    both<4096, 64>(mats, NM, sink);  both<4096, 128>(mats, NM, sink); both<4096, 256>(mats, NM, sink);
    both<8192, 128>(mats, NM, sink); both<8192, 256>(mats, NM, sink); both<8192, 512>(mats, NM, sink);
    both<16384, 256>(mats, NM, sink); both<16384, 512>(mats, NM, sink); both<16384, 1024>(mats, NM, sink);
    run<4096, 128, true, 16, true>(mats, NM, sink); run<16384, 512, true, 16, true>(mats, NM, sink);
    run<4096, 128, true, 4, true>(mats, NM, sink);  run<16384, 512, true, 4, true>(mats, NM, sink);
    return 0;
}
