// Dev probe: how fast does HBM take a COLD [8192][16 KiB] matrix of bf16 outputs when it is written in GEMM tiles -- every
// workgroup (4 waves) owns a TM x TN tile and each wave store instruction covers R rows x (1024 / R) bytes -- against a plain
// contiguous stream?  Eight 128 MiB matrices are rotated so that no launch finds its output in the 256 MiB Infinity Cache.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int ROWS = 8192, ROW_BYTES = 16384;

// tile TM rows x TNB bytes; a wave instruction writes R rows x (1024 / R) contiguous bytes; waves split the tile's rows
template <int TM, int TNB, int R, bool NT>
__global__ void __launch_bounds__(256) wr(char *__restrict__ mat, float seed)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    constexpr int tiles_n = ROW_BYTES / TNB;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    char *base = mat + (size_t)tm * TM * ROW_BYTES + (size_t)tn * TNB;
    constexpr int SEG = 1024 / R;                 // contiguous bytes per row and instruction
    constexpr int LPR = SEG / 16;                 // lanes per row
    const f32x4 v = {seed, seed, seed, seed};
    // wave w owns rows w * TM/4 .. ; walks its (rows x TNB) block in instructions of R rows x SEG bytes
    for (int r0 = 0; r0 < TM / 4; r0 += R)
        for (int c0 = 0; c0 < TNB; c0 += SEG) {
            char *p = base + (size_t)(w * (TM / 4) + r0 + lane / LPR) * ROW_BYTES + c0 + (lane % LPR) * 16;
            if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(p)); else *reinterpret_cast<f32x4 *>(p) = v;
        }
}
__global__ void __launch_bounds__(256) wr_stream(char *__restrict__ mat, float seed)
{
    const f32x4 v = {seed, seed, seed, seed};
    char *base = mat + (size_t)blockIdx.x * 32768;
    for (int i = 0; i < 8; ++i) __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(base + i * 4096 + threadIdx.x * 16));
}

template <typename F> void timeit(const char *name, F launch, char **mats, int nm)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    std::vector<float> ms;
    for (int i = 0; i < 28; ++i) {
        hipEventRecord(a); launch(mats[i % nm]); hipEventRecord(b); hipEventSynchronize(b);
        float t; hipEventElapsedTime(&t, a, b);
        if (i >= 4) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    const double bytes = (double)ROWS * ROW_BYTES;
    printf("%-44s median %7.1f GB/s (%6.1f us)  best %7.1f GB/s\n", name, bytes / ms[ms.size() / 2] / 1e6, ms[ms.size() / 2] * 1e3, bytes / ms[0] / 1e6);
}
#define RUN(TM, TNB, R, NT) timeit("tile " #TM " x " #TNB " B, " #R " rows/instr" #NT, [](char *m) { wr<TM, TNB, R, NT><<<(ROWS / TM) * (ROW_BYTES / TNB), 256>>>(m, 1.f); }, mats, NM)

int main()
{
    constexpr int NM = 8;
    char *mats[NM];
    for (int i = 0; i < NM; ++i) { hipMalloc(&mats[i], (size_t)ROWS * ROW_BYTES); hipMemset(mats[i], 0, (size_t)ROWS * ROW_BYTES); }
    timeit("contiguous stream (32 KiB per workgroup), nt", [](char *m) { wr_stream<<<ROWS * ROW_BYTES / 32768, 256>>>(m, 1.f); }, mats, NM);
    RUN(128, 256, 8, true);   RUN(128, 256, 8, false);  RUN(128, 256, 4, true);
    RUN(128, 512, 8, true);   RUN(128, 512, 2, true);   RUN(64, 512, 2, true);
    RUN(128, 1024, 1, true);  RUN(256, 512, 2, true);   RUN(64, 1024, 1, true);  RUN(32, 2048, 1, true);
    printf("-- warm (one matrix)\n");
    timeit("contiguous stream, nt", [](char *m) { wr_stream<<<ROWS * ROW_BYTES / 32768, 256>>>(m, 1.f); }, mats, 1);
    timeit("tile 128 x 256 B, 8 rows/instr, nt", [](char *m) { wr<128, 256, 8, true><<<(ROWS / 128) * (ROW_BYTES / 256), 256>>>(m, 1.f); }, mats, 1);
    return 0;
}
