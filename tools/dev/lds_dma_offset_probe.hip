// Dev probe: does the instruction offset of global_load_lds_dwordx4 move the LDS destination as well as the global source?
// (M0 = LDS base; one wave; source dword i holds i.)   build: hipcc --offload-arch=gfx950 -o /tmp/ldsdma tools/dev/lds_dma_offset_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void probe(const uint32_t *src, uint32_t *out)
{
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    uint32_t *l = reinterpret_cast<uint32_t *>(smem);
    for (int i = threadIdx.x; i < 4096; i += 64) l[i] = 0xffffffffu;
    __syncthreads();
    const uint32_t voff = threadIdx.x * 16;
    const uint32_t ldsb = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char *)smem;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:2048\n\ts_waitcnt vmcnt(0)" ::"v"(voff), "s"(src), "s"(ldsb) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += 64) out[i] = l[i];
}
int main()
{
    std::vector<uint32_t> h(8192);
    for (int i = 0; i < 8192; ++i) h[i] = i;
    uint32_t *s, *o;
    hipMalloc(&s, 8192 * 4); hipMalloc(&o, 4096 * 4);
    hipMemcpy(s, h.data(), 8192 * 4, hipMemcpyHostToDevice);
    probe<<<1, 64, 16384>>>(s, o);
    std::vector<uint32_t> r(4096);
    hipMemcpy(r.data(), o, 4096 * 4, hipMemcpyDeviceToHost);
    int first = -1, n = 0;
    for (int i = 0; i < 4096; ++i) if (r[i] != 0xffffffffu) { if (first < 0) first = i; ++n; }
    printf("offset:2048 -> %d dwords written, first LDS dword %d (byte %d) holds source dword %u (byte %u)\n", n, first, first * 4, first >= 0 ? r[first] : 0, first >= 0 ? r[first] * 4 : 0);
    return 0;
}
