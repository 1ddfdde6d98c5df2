// gemm_add.hip -- the "+ C" of cmma::execute(a, b, c, d) (crates/cubecl-core/src/frontend/cmma.rs:1066-1110) at tensor
// level: D = A * B + C.  mi355_gemm_add (gemm.cpp) lets the selected GEMM kernel write the f32 product into library
// scratch and this kernel forms product + C in f32 and rounds ONCE to the output type -- the accumulator fragment of the
// reference is f32, so no intermediate 16-bit rounding may happen.  HBM-bound: per element 4 (product) + 2 x sizeof(C) bytes.
#include "gemm_common.hpp"

using namespace mi355;

namespace {

template <int DT_C>
__device__ __forceinline__ float widen(const void *p, int64_t i)
{
    if (DT_C == MI355_DTYPE_F32) return static_cast<const float *>(p)[i];
    const uint16_t bits = static_cast<const uint16_t *>(p)[i];
    if (DT_C == MI355_DTYPE_BF16) return __uint_as_float((uint32_t)bits << 16);
    return (float)__builtin_bit_cast(_Float16, bits);
}

// One thread per 4 consecutive columns of one row; rows / batches walk the grid's y / z.
template <int DT_C, bool VEC>
__global__ void __launch_bounds__(256)
add_c_kernel(const float *__restrict__ prod, const void *c_in, void *d_out, int64_t m, int64_t n, int64_t batch, int64_t ldc,
             int64_t stride_c)
{
    typedef uint16_t u16x4 __attribute__((ext_vector_type(4)));
    for (int64_t b = blockIdx.z; b < batch; b += gridDim.z)
        for (int64_t row = blockIdx.y; row < m; row += gridDim.y) {
            const float *p = prod + (b * m + row) * n;
            const int64_t base = b * stride_c + row * ldc;
            if constexpr (VEC) {
                for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n / 4; q += (int64_t)gridDim.x * 256) {
                    f32x4 acc = *reinterpret_cast<const f32x4 *>(p + q * 4);
                    if constexpr (DT_C == MI355_DTYPE_F32) {
                        acc += *reinterpret_cast<const f32x4 *>(static_cast<const float *>(c_in) + base + q * 4);
                        *reinterpret_cast<f32x4 *>(static_cast<float *>(d_out) + base + q * 4) = acc;
                    } else {
                        const u16x4 cv = *reinterpret_cast<const u16x4 *>(static_cast<const uint16_t *>(c_in) + base + q * 4);
                        u16x4 o;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float c = DT_C == MI355_DTYPE_BF16 ? __uint_as_float((uint32_t)cv[r] << 16)
                                                                      : (float)__builtin_bit_cast(_Float16, (uint16_t)cv[r]);
                            o[r] = f32_to_lp<DT_C>(acc[r] + c);
                        }
                        *reinterpret_cast<u16x4 *>(static_cast<uint16_t *>(d_out) + base + q * 4) = o;
                    }
                }
            } else {
                for (int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x; col < n; col += (int64_t)gridDim.x * 256) {
                    const float v = p[col] + widen<DT_C>(c_in, base + col);
                    if (DT_C == MI355_DTYPE_F32) static_cast<float *>(d_out)[base + col] = v;
                    else static_cast<uint16_t *>(d_out)[base + col] = f32_to_lp<DT_C>(v);
                }
            }
        }
}

}  // namespace

namespace mi355 {

void launch_add_c(hipStream_t s, const float *prod, const void *c_in, void *d_out, int64_t batch, int64_t m, int64_t n,
                  int32_t dtype_c, int64_t ldc, int64_t stride_c)
{
    const int64_t csz = dtype_c == MI355_DTYPE_F32 ? 4 : 2;
    const bool vec = n % 4 == 0 && ldc % 4 == 0 && stride_c % 4 == 0 &&
                     ((reinterpret_cast<uintptr_t>(c_in) | reinterpret_cast<uintptr_t>(d_out)) % (uintptr_t)(4 * csz)) == 0;
    const int64_t per_row = vec ? n / 4 : n;
    const dim3 grid((uint32_t)std::max<int64_t>(1, std::min<int64_t>((per_row + 255) / 256, 64)),
                    (uint32_t)std::min<int64_t>(m, 65535), (uint32_t)std::min<int64_t>(batch, 1024));
#define ADD(DT)                                                                                                                   \
    do {                                                                                                                          \
        if (vec) hipLaunchKernelGGL((add_c_kernel<DT, true>), grid, dim3(256), 0, s, prod, c_in, d_out, m, n, batch, ldc, stride_c);   \
        else hipLaunchKernelGGL((add_c_kernel<DT, false>), grid, dim3(256), 0, s, prod, c_in, d_out, m, n, batch, ldc, stride_c);      \
    } while (0)
    if (dtype_c == MI355_DTYPE_F32) ADD(MI355_DTYPE_F32);
    else if (dtype_c == MI355_DTYPE_BF16) ADD(MI355_DTYPE_BF16);
    else ADD(MI355_DTYPE_F16);
#undef ADD
}

}  // namespace mi355
