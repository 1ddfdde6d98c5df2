// gemm_relayout.hip -- operand re-layout in front of the MFMA kernels + library-owned scratch.
//
// The MFMA kernels want both operands K-contiguous (A [M][K], B [N][K]): an MFMA fragment is 8 (16-bit) or 4
// (f32) consecutive k-values of one row, and the LDS-DMA image of a K-tile is lane-linear, so a row-major B
// [K][N] or a transposed A [K][M] cannot be staged by DMA directly.  The reference's launchers do the same thing
// one level up: `matrix_batch_layout` (crates/cubecl-std/src/tensor/matrix_batch_layout.rs:21-79) classifies the
// operand and `into_contiguous` re-lays it out before the kernel when the layout is not the kernel's.  Here the
// re-layout is a 64 x 64-tile transpose through LDS into library-owned, per-stream scratch.
//
// Roofline: HBM (read + write the operand once: 2 x rows x cols x sizeof).  For an 8192^3 bf16 GEMM with row-major
// B that is 256 MB of extra traffic = ~50 us in front of a ~760 us kernel (measured: 3.x TB/s);
// against the generic scalar-FMA kernel such a GEMM would otherwise fall to, it is two orders of magnitude.
#include <algorithm>

#include "gemm_common.hpp"

using namespace mi355;

namespace {

// dst[c][r] = src[r][c] for a rows x cols matrix (per batch entry).  64 x 64 tile per workgroup of 256 threads;
// 16-byte accesses on both sides when the tile is interior and everything is 16-byte aligned, element-wise at
// the ragged edges.  T = uint8_t (fp8 bits), uint16_t (bf16 / f16 bits) or uint32_t (f32 bits).
template <typename T>
__global__ void __launch_bounds__(256)
transpose_kernel(const T *__restrict__ src, T *__restrict__ dst, int64_t rows, int64_t cols, int64_t ld_src, int64_t ld_dst,
                 int64_t stride_src, int64_t stride_dst, uint32_t tiles_c, int vec_ok)
{
    constexpr int TS = 64;
    constexpr int VE = 16 / sizeof(T);                 // elements per 16-byte access: 8 / 4
    __shared__ T tile[TS][TS + VE + 1];                 // odd-ish pitch: column reads spread over the banks
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)(blockIdx.x / tiles_c) * TS, c0 = (int64_t)(blockIdx.x % tiles_c) * TS;
    const T *s = src + (int64_t)blockIdx.y * stride_src;
    T *d = dst + (int64_t)blockIdx.y * stride_dst;
    const bool interior = vec_ok && (r0 + TS <= rows) && (c0 + TS <= cols);
    constexpr int TPR = TS / VE;                        // threads per tile row when vectorised: 8 / 16
    if (interior) {
        typedef T vec __attribute__((ext_vector_type(VE)));
#pragma unroll
        for (int p = 0; p < TS * TPR / 256; ++p) {
            const int lin = tid + p * 256, r = lin / TPR, cv = lin % TPR;
            const vec v = *reinterpret_cast<const vec *>(s + (r0 + r) * ld_src + c0 + cv * VE);
#pragma unroll
            for (int e = 0; e < VE; ++e) tile[r][cv * VE + e] = v[e];
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < TS * TPR / 256; ++p) {
            const int lin = tid + p * 256, c = lin / TPR, rv = lin % TPR;        // output row c, 8 (4) source rows
            vec v;
#pragma unroll
            for (int e = 0; e < VE; ++e) v[e] = tile[rv * VE + e][c];
            *reinterpret_cast<vec *>(d + (c0 + c) * ld_dst + r0 + rv * VE) = v;
        }
    } else {
        for (int lin = tid; lin < TS * TS; lin += 256) {
            const int r = lin / TS, c = lin % TS;
            if (r0 + r < rows && c0 + c < cols) tile[r][c] = s[(r0 + r) * ld_src + c0 + c];
        }
        __syncthreads();
        for (int lin = tid; lin < TS * TS; lin += 256) {
            const int c = lin / TS, r = lin % TS;
            if (r0 + r < rows && c0 + c < cols) d[(c0 + c) * ld_dst + r0 + r] = tile[r][c];
        }
    }
}

// dst[r][0..cols) = src[r][0..cols), dst[r][cols..cols_pad) = 0: gives ragged-K or unaligned operands the
// K-tile-multiple, 16-byte-aligned rows the MFMA kernels need (zero k-columns add nothing to the products).
template <typename T>
__global__ void __launch_bounds__(256)
pad_copy_kernel(const T *__restrict__ src, T *__restrict__ dst, int64_t rows, int64_t cols, int64_t cols_pad, int64_t ld_src,
                int64_t ld_dst, int64_t stride_src, int64_t stride_dst, int vec_ok)
{
    constexpr int VE = 16 / sizeof(T);
    typedef T vec __attribute__((ext_vector_type(VE)));
    const T *s = src + (int64_t)blockIdx.z * stride_src;
    T *d = dst + (int64_t)blockIdx.z * stride_dst;
    for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) {
        const T *sr = s + r * ld_src;
        T *dr = d + r * ld_dst;
        const int64_t full = vec_ok ? cols / VE : 0;                   // whole 16-byte pieces of valid data
        for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < cols_pad / VE; q += (int64_t)gridDim.x * 256) {
            vec v;
            if (q < full) v = *reinterpret_cast<const vec *>(sr + q * VE);
            else {
#pragma unroll
                for (int e = 0; e < VE; ++e) v[e] = (q * VE + e < cols) ? sr[q * VE + e] : (T)0;
            }
            *reinterpret_cast<vec *>(dr + q * VE) = v;
        }
    }
}

}  // namespace

namespace mi355 {

void launch_pad_copy(hipStream_t s, const void *src, void *dst, int64_t rows, int64_t cols, int64_t cols_pad, int64_t ld_src,
                     int64_t ld_dst, int64_t batch, int64_t stride_src, int64_t stride_dst, int esz)
{
    if (rows <= 0 || cols_pad <= 0 || batch <= 0) return;
    const int64_t ve = 16 / esz;
    const int vec_ok = (reinterpret_cast<uintptr_t>(src) & 15u) == 0 && (ld_src % ve) == 0 && (stride_src % ve) == 0;
    const dim3 grid((uint32_t)std::max<int64_t>(1, std::min<int64_t>((cols_pad / ve + 255) / 256, 16)),
                    (uint32_t)std::min<int64_t>(rows, 4096), (uint32_t)batch);
    if (esz == 1)
        hipLaunchKernelGGL(pad_copy_kernel<uint8_t>, grid, dim3(256), 0, s, static_cast<const uint8_t *>(src),
                           static_cast<uint8_t *>(dst), rows, cols, cols_pad, ld_src, ld_dst, stride_src, stride_dst, vec_ok);
    else if (esz == 2)
        hipLaunchKernelGGL(pad_copy_kernel<uint16_t>, grid, dim3(256), 0, s, static_cast<const uint16_t *>(src),
                           static_cast<uint16_t *>(dst), rows, cols, cols_pad, ld_src, ld_dst, stride_src, stride_dst, vec_ok);
    else
        hipLaunchKernelGGL(pad_copy_kernel<uint32_t>, grid, dim3(256), 0, s, static_cast<const uint32_t *>(src),
                           static_cast<uint32_t *>(dst), rows, cols, cols_pad, ld_src, ld_dst, stride_src, stride_dst, vec_ok);
}

void launch_transpose(hipStream_t s, const void *src, void *dst, int64_t rows, int64_t cols, int64_t ld_src, int64_t ld_dst,
                      int64_t batch, int64_t stride_src, int64_t stride_dst, int esz)
{
    if (rows <= 0 || cols <= 0 || batch <= 0) return;
    const uint32_t tiles_r = (uint32_t)((rows + 63) / 64), tiles_c = (uint32_t)((cols + 63) / 64);
    const dim3 grid(tiles_r * tiles_c, (uint32_t)batch);
    const int64_t ve = 16 / esz;
    const int vec_ok = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0 && (ld_src % ve) == 0 &&
                       (ld_dst % ve) == 0 && (stride_src % ve) == 0 && (stride_dst % ve) == 0;
    if (esz == 1)
        hipLaunchKernelGGL(transpose_kernel<uint8_t>, grid, dim3(256), 0, s, static_cast<const uint8_t *>(src),
                           static_cast<uint8_t *>(dst), rows, cols, ld_src, ld_dst, stride_src, stride_dst, tiles_c, vec_ok);
    else if (esz == 2)
        hipLaunchKernelGGL(transpose_kernel<uint16_t>, grid, dim3(256), 0, s, static_cast<const uint16_t *>(src),
                           static_cast<uint16_t *>(dst), rows, cols, ld_src, ld_dst, stride_src, stride_dst, tiles_c, vec_ok);
    else
        hipLaunchKernelGGL(transpose_kernel<uint32_t>, grid, dim3(256), 0, s, static_cast<const uint32_t *>(src),
                           static_cast<uint32_t *>(dst), rows, cols, ld_src, ld_dst, stride_src, stride_dst, tiles_c, vec_ok);
}

}  // namespace mi355
