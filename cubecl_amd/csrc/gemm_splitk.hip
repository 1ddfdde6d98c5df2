// gemm_splitk.hip -- split-K support for the skinny bf16/f16 GEMM shapes: library-owned f32 slab scratch and
// the fold kernel.
//
// Roofline: HBM.  A skinny GEMM (M or N << 256 tiles' worth, K large) is a streaming problem: its time is
// the operand bytes over HBM bandwidth, provided enough workgroups stream at once.  With 128x128 tiles a
// 64 x 8192 x 8192 product has 64 tiles for 256 CUs, so K is cut into `splits` slices (gemm_lp128.hip,
// blockIdx.z); each slice writes an f32 partial slab [batch][M][N]; this kernel adds the slabs IN SLICE
// ORDER (deterministic, no float atomics) and converts to the output type.  Algorithmic bytes of the fold:
// splits x M x N x 4 read + M x N x sizeof(C) written.
#include <algorithm>

#include "gemm_common.hpp"

using namespace mi355;

namespace {

// One workgroup = 64 vec4 outputs (256 consecutive floats of the flattened [batch][M][N] slab: 1 KiB per slice, whole lines)
// x 4 waves; wave g adds the slices s = g, g + 4, g + 8, ... (at most 8 of the <= 32: all its loads are issued before the
// first add), the four partial sums meet in LDS and are added in wave order.  Deterministic: the summation tree depends on
// `splits` and the output size only.  SMALL outputs (up to ~100 k vec4) take this form since round 3: the serial form below ran
// one block per output ROW -- a 128 x 256 output (32 slices) kept 128 waves busy with 32 serial loads each and took 11 us, more
// than the GEMM launch it followed (rocprofv3: fold 11.0 us average against 10.0 us for gemm_lp128 over 128 x 256 x 8192,
// 512 x 1024 x 2048 and 96 x 96 x 16384).  Whole GEMM, interleaved: 128 x 256 x 8192 22.2 -> 11.0 us, 96 x 96 x 16384 23.0 -> 12.9,
// 512 x 512 x 8192 20.7 -> 16.5.
template <int DT_C>
__global__ void __launch_bounds__(256)
splitk_fold_kernel(const float *__restrict__ slabs, uint32_t splits, int64_t slab_stride, int64_t m, int64_t n,
                   void *__restrict__ c, int64_t ldc, int64_t stride_c, int aligned)
{
    __shared__ f32x4 part[4][64];
    const uint32_t lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const uint32_t n4 = (uint32_t)(n / 4);                                   // (n is a multiple of 4 and m * n / 4 < 2^32: checked by the launcher)
    const uint32_t v = blockIdx.x * 64 + lane;                               // vec4 index inside one batch entry's [M][N] slab
    const int64_t b = blockIdx.z;
    const bool live = v < (uint32_t)m * n4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        const float *p = slabs + b * m * n + (int64_t)v * 4;
        for (uint32_t s0 = g; s0 < splits; s0 += 16) {                       // at most two rounds of four loads in flight
            f32x4 t[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t sl = s0 + 4 * i;
                t[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (sl < splits) t[i] = *reinterpret_cast<const f32x4 *>(p + (int64_t)sl * slab_stride);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) acc += t[i];
        }
    }
    part[g][lane] = acc;
    __syncthreads();
    if (g != 0 || !live) return;
    acc = part[0][lane] + part[1][lane];
    acc += part[2][lane];
    acc += part[3][lane];
    const uint32_t row = v / n4, q = v - row * n4;
    const int64_t idx = b * stride_c + (int64_t)row * ldc + (int64_t)q * 4;
    // `aligned`: every row of C starts on a 16- / 8-byte boundary (launcher) -- one vector store; otherwise the SAME sum leaves
    // element by element: which summation tree an output gets must depend on (m, n, splits) only, never on where C happens
    // to sit (advisor, round 3: a misaligned or pitched C used to take the serial form and could differ in the last bit)
    if (DT_C == MI355_DTYPE_F32) {
        float *dst = static_cast<float *>(c) + idx;
        if (aligned) *reinterpret_cast<f32x4 *>(dst) = acc;
        else {
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[r] = acc[r];
        }
    } else {
        uint16_t *dst = static_cast<uint16_t *>(c) + idx;
        if (aligned) {
            u32x2 o = {f32x2_to_lp<DT_C>(acc[0], acc[1]), f32x2_to_lp<DT_C>(acc[2], acc[3])};
            *reinterpret_cast<u32x2 *>(dst) = o;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[r] = f32_to_lp<DT_C>(acc[r]);
        }
    }
}

// Large outputs: one thread per 4 consecutive columns of one row, the slices added serially -- enough threads are in flight to
// hide the latency, and nothing but the loads and the store is executed (from ~100 k vec4 outputs up this form is 2-5 % ahead of
// the one above; below, up to 2 x behind: interleaved, profiles/r03_split_k_fold.txt).
template <int DT_C>
__global__ void __launch_bounds__(256)
splitk_fold_rows_kernel(const float *__restrict__ slabs, uint32_t splits, int64_t slab_stride, int64_t m, int64_t n,
                        void *__restrict__ c, int64_t ldc, int64_t stride_c)
{
    const int64_t n4 = n / 4;
    const int64_t row = blockIdx.y;
    const int64_t b = blockIdx.z;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n4; q += (int64_t)gridDim.x * 256) {
        const float *p = slabs + (b * m + row) * n + q * 4;
        f32x4 acc = *reinterpret_cast<const f32x4 *>(p);
        for (uint32_t s = 1; s < splits; ++s) acc += *reinterpret_cast<const f32x4 *>(p + (int64_t)s * slab_stride);
        const int64_t idx = b * stride_c + row * ldc + q * 4;
        if (DT_C == MI355_DTYPE_F32) {
            float *dst = static_cast<float *>(c) + idx;
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[r] = acc[r];
        } else {
            uint16_t *dst = static_cast<uint16_t *>(c) + idx;
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[r] = f32_to_lp<DT_C>(acc[r]);
        }
    }
}

// ragged N (not a multiple of 4): scalar version
template <int DT_C>
__global__ void __launch_bounds__(256)
splitk_fold_scalar_kernel(const float *__restrict__ slabs, uint32_t splits, int64_t slab_stride, int64_t m, int64_t n,
                          void *__restrict__ c, int64_t ldc, int64_t stride_c)
{
    const int64_t row = blockIdx.y, b = blockIdx.z;
    for (int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x; col < n; col += (int64_t)gridDim.x * 256) {
        const float *p = slabs + (b * m + row) * n + col;
        float acc = p[0];
        for (uint32_t s = 1; s < splits; ++s) acc += p[(int64_t)s * slab_stride];
        const int64_t idx = b * stride_c + row * ldc + col;
        if (DT_C == MI355_DTYPE_F32) static_cast<float *>(c)[idx] = acc;
        else static_cast<uint16_t *>(c)[idx] = f32_to_lp<DT_C>(acc);
    }
}

}  // namespace

namespace mi355 {

int32_t splitk_scratch(mi355_ctx *ctx, hipStream_t s, size_t bytes, float **out)
{
    void *p = nullptr;
    const int32_t rc = scratch_get(ctx, s, SCRATCH_SPLITK, bytes, &p);   // a failure makes the caller fall back to the unsplit kernel
    *out = static_cast<float *>(p);
    return rc;
}

void launch_splitk_fold(hipStream_t s, const float *slabs, uint32_t splits, int64_t slab_stride, int64_t batch, int64_t m,
                        int64_t n, void *c, int32_t dtype_c, int64_t ldc, int64_t stride_c)
{
    const bool vec = (n % 4) == 0;
    // small outputs: 64 vec4 x 4 slice groups per workgroup, 16- / 8-byte stores (rows of C must start on that boundary)
    const int64_t csz = dtype_c == MI355_DTYPE_F32 ? 4 : 2;
    const bool wide = vec && splits <= 32 && m * (n / 4) <= 98304;          // the form -- and with it the summation tree -- follows (m, n, splits) only
    const int aligned = ((ldc % 4) == 0 && (stride_c % 4) == 0 && (reinterpret_cast<uintptr_t>(c) % (4 * csz)) == 0) ? 1 : 0;
    const int64_t work = vec ? n / 4 : n;
    dim3 grid((uint32_t)std::max<int64_t>(1, std::min<int64_t>((work + 255) / 256, 64)), (uint32_t)m, (uint32_t)batch);
    if (wide) grid = dim3((uint32_t)((m * (n / 4) + 63) / 64), 1, (uint32_t)batch);
#define FOLD(K, DT) hipLaunchKernelGGL((K<DT>), grid, dim3(256), 0, s, slabs, splits, slab_stride, m, n, c, ldc, stride_c)
#define FOLDW(DT) hipLaunchKernelGGL((splitk_fold_kernel<DT>), grid, dim3(256), 0, s, slabs, splits, slab_stride, m, n, c, ldc, stride_c, aligned)
    if (wide) {
        if (dtype_c == MI355_DTYPE_F32) FOLDW(MI355_DTYPE_F32);
        else if (dtype_c == MI355_DTYPE_BF16) FOLDW(MI355_DTYPE_BF16);
        else FOLDW(MI355_DTYPE_F16);
    } else if (vec) {
        if (dtype_c == MI355_DTYPE_F32) FOLD(splitk_fold_rows_kernel, MI355_DTYPE_F32);
        else if (dtype_c == MI355_DTYPE_BF16) FOLD(splitk_fold_rows_kernel, MI355_DTYPE_BF16);
        else FOLD(splitk_fold_rows_kernel, MI355_DTYPE_F16);
    } else {
        if (dtype_c == MI355_DTYPE_F32) FOLD(splitk_fold_scalar_kernel, MI355_DTYPE_F32);
        else if (dtype_c == MI355_DTYPE_BF16) FOLD(splitk_fold_scalar_kernel, MI355_DTYPE_BF16);
        else FOLD(splitk_fold_scalar_kernel, MI355_DTYPE_F16);
    }
#undef FOLD
#undef FOLDW
}

}  // namespace mi355
