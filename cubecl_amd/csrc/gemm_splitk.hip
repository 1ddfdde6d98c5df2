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

template <int DT_C>
__global__ void __launch_bounds__(256)
splitk_fold_kernel(const float *__restrict__ slabs, uint32_t splits, int64_t slab_stride, int64_t m, int64_t n,
                   void *__restrict__ c, int64_t ldc, int64_t stride_c)
{
    // one thread per 4 consecutive columns of one row (n is a multiple of 4 here: checked by the launcher)
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
    const int64_t work = vec ? n / 4 : n;
    const dim3 grid((uint32_t)std::max<int64_t>(1, std::min<int64_t>((work + 255) / 256, 64)), (uint32_t)m, (uint32_t)batch);
#define FOLD(K, DT) hipLaunchKernelGGL((K<DT>), grid, dim3(256), 0, s, slabs, splits, slab_stride, m, n, c, ldc, stride_c)
    if (vec) {
        if (dtype_c == MI355_DTYPE_F32) FOLD(splitk_fold_kernel, MI355_DTYPE_F32);
        else if (dtype_c == MI355_DTYPE_BF16) FOLD(splitk_fold_kernel, MI355_DTYPE_BF16);
        else FOLD(splitk_fold_kernel, MI355_DTYPE_F16);
    } else {
        if (dtype_c == MI355_DTYPE_F32) FOLD(splitk_fold_scalar_kernel, MI355_DTYPE_F32);
        else if (dtype_c == MI355_DTYPE_BF16) FOLD(splitk_fold_scalar_kernel, MI355_DTYPE_BF16);
        else FOLD(splitk_fold_scalar_kernel, MI355_DTYPE_F16);
    }
#undef FOLD
}

}  // namespace mi355
