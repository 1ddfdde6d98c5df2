// Dev sweep: 1 GiB f32 streaming-read variants (sum into registers, guarded store) to find the access
// shape with the highest HBM read rate on MI355X.  Prints GB/s per variant (best and median of 20).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int BLOCK, int UNROLL, int LOADK, bool CONTIG>
__global__ void __launch_bounds__(BLOCK) rd(const f32x4 *__restrict__ buf, uint64_t nvec, float *sink)
{
    f32x4 acc[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc[u] = (f32x4){0, 0, 0, 0};
    const uint64_t tile = (uint64_t)BLOCK * UNROLL, tiles = nvec / tile;
    uint64_t t0, t1, step;
    if (CONTIG) { const uint64_t per = (tiles + gridDim.x - 1) / gridDim.x; t0 = blockIdx.x * per; t1 = std::min<uint64_t>(tiles, t0 + per); step = 1; }
    else { t0 = blockIdx.x; t1 = tiles; step = gridDim.x; }
    for (uint64_t t = t0; t < t1; t += step) {
        const f32x4 *p = buf + t * tile + threadIdx.x;
        f32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (LOADK == 0) v[u] = p[(uint64_t)u * BLOCK];
            else if (LOADK == 1) v[u] = __builtin_nontemporal_load(p + (uint64_t)u * BLOCK);
            else asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v[u]) : "v"(p + (uint64_t)u * BLOCK) : "memory");
        }
        if (LOADK == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc[u] += v[u];
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int u = 1; u < UNROLL; ++u) s += acc[u];
    if (s[0] + s[1] + s[2] + s[3] == 1.2345e38f) sink[0] = s[0];
}

template <int BLOCK, int UNROLL, int LOADK, bool CONTIG>
void run(const f32x4 *buf, uint64_t nvec, float *sink, int grid)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    std::vector<float> ms;
    for (int i = 0; i < 23; ++i) {
        hipEventRecord(a);
        rd<BLOCK, UNROLL, LOADK, CONTIG><<<grid, BLOCK>>>(buf, nvec, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float t; hipEventElapsedTime(&t, a, b);
        if (i >= 3) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    const double bytes = (double)nvec * 16;
    printf("block %4d unroll %2d load %s %s grid %5d: best %7.1f GB/s  median %7.1f GB/s\n", BLOCK, UNROLL,
           LOADK == 0 ? "plain" : LOADK == 1 ? "nt   " : "sc0sc1", CONTIG ? "contig " : "strided", grid, bytes / ms[0] / 1e6, bytes / ms[ms.size() / 2] / 1e6);
}

int main()
{
    const uint64_t bytes = 1ull << 30, nvec = bytes / 16;
    f32x4 *buf; float *sink; hipMalloc(&buf, bytes); hipMemset(buf, 0, bytes); hipMalloc(&sink, 64);
    for (int g : {512, 1024, 2048, 4096, 8192}) {
        run<256, 8, 1, false>(buf, nvec, sink, g);
        run<256, 8, 0, false>(buf, nvec, sink, g);
    }
    for (int g : {1024, 2048, 4096}) {
        run<256, 4, 1, false>(buf, nvec, sink, g);
        run<256, 16, 1, false>(buf, nvec, sink, g);
        run<512, 8, 1, false>(buf, nvec, sink, g);
        run<1024, 4, 1, false>(buf, nvec, sink, g);
        run<256, 8, 1, true>(buf, nvec, sink, g);
        run<256, 8, 2, false>(buf, nvec, sink, g);
    }
    run<256, 8, 1, true>(buf, nvec, sink, 256);
    run<512, 8, 1, true>(buf, nvec, sink, 256);
    run<1024, 8, 1, true>(buf, nvec, sink, 256);
    run<1024, 4, 1, false>(buf, nvec, sink, 256);
    run<1024, 8, 1, false>(buf, nvec, sink, 512);
    return 0;
}
