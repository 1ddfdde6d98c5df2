// fill.hip -- on-device synthetic data (counter-based RNG shared bit-for-bit with
// oracle/oracle.c) and whole-buffer dtype casts.  Inputs of the benches are generated in HBM so
// the timed region never includes PCIe ("uploads dominate unless inputs
// are generated on-device").
#include "internal.hpp"
#include "fp8.hpp"

#include <hip/hip_fp16.h>

#include <algorithm>

using namespace mi355;

namespace {

__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__device__ __forceinline__ float rng_value(uint64_t key, uint64_t i, float lo, float scale)
{
    const uint64_t h = splitmix64(key + i);
    const float u = (float)(uint32_t)(h >> 40) * (1.0f / 16777216.0f);
    return __fmaf_rn(scale, u, lo);
}

// round-to-nearest-even f32 -> bf16, same integer recipe as oracle_f32_to_bf16
__device__ __forceinline__ uint16_t f32_to_bf16(float f)
{
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x0040u);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

__device__ __forceinline__ uint16_t f32_to_f16(float f)
{
    const __half h = __float2half_rn(f);
    return __half_as_ushort(h);
}

template <int DTYPE>
__global__ void __launch_bounds__(256) fill_uniform_kernel(void *__restrict__ dst, uint64_t n, uint64_t key, float lo,
                                                           float scale)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float v = rng_value(key, i, lo, scale);
        if (DTYPE == MI355_DTYPE_F32) static_cast<float *>(dst)[i] = v;
        else if (DTYPE == MI355_DTYPE_U8) static_cast<uint8_t *>(dst)[i] = (uint8_t)(int)floorf(v);   // bytes: floor of the value
        else if (DTYPE == MI355_DTYPE_BF16) static_cast<uint16_t *>(dst)[i] = f32_to_bf16(v);
        else if (DTYPE == MI355_DTYPE_F8E4M3) static_cast<uint8_t *>(dst)[i] = f32_to_e4m3(v);
        else if (DTYPE == MI355_DTYPE_F8E5M2) static_cast<uint8_t *>(dst)[i] = f32_to_e5m2(v);
        else static_cast<uint16_t *>(dst)[i] = f32_to_f16(v);
    }
}

template <int SRC, int DST>
__global__ void __launch_bounds__(256) cast_kernel(const void *__restrict__ src, void *__restrict__ dst, uint64_t n)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float v;
        if (SRC == MI355_DTYPE_F32) v = static_cast<const float *>(src)[i];
        else if (SRC == MI355_DTYPE_BF16) v = __uint_as_float((uint32_t)static_cast<const uint16_t *>(src)[i] << 16);
        else if (SRC == MI355_DTYPE_F8E4M3) v = e4m3_to_f32(static_cast<const uint8_t *>(src)[i]);
        else if (SRC == MI355_DTYPE_F8E5M2) v = e5m2_to_f32(static_cast<const uint8_t *>(src)[i]);
        else v = __half2float(__ushort_as_half(static_cast<const uint16_t *>(src)[i]));
        if (DST == MI355_DTYPE_F32) static_cast<float *>(dst)[i] = v;
        else if (DST == MI355_DTYPE_BF16) static_cast<uint16_t *>(dst)[i] = f32_to_bf16(v);
        else if (DST == MI355_DTYPE_F8E4M3) static_cast<uint8_t *>(dst)[i] = f32_to_e4m3(v);
        else if (DST == MI355_DTYPE_F8E5M2) static_cast<uint8_t *>(dst)[i] = f32_to_e5m2(v);
        else static_cast<uint16_t *>(dst)[i] = f32_to_f16(v);
    }
}

uint32_t grid_for(const mi355_ctx *ctx, uint64_t n)
{
    const uint64_t blocks = (n + 255) / 256;
    return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(blocks, (uint64_t)ctx->props.num_streaming_multiprocessors * 16));
}

uint64_t host_splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// Identity matrix (crates/cubecl-std/src/tensor/identity.rs:8-34): out[r][c] = (r == c), rows `ld` elements apart.
// T carries the element's bits; `one` is the dtype's representation of 1.
template <typename T>
__global__ void __launch_bounds__(256) identity_kernel(T *__restrict__ out, uint64_t dim, uint64_t ld, T one)
{
    const uint64_t total = dim * dim, stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const uint64_t r = i / dim, c = i - r * dim;
        out[r * ld + c] = r == c ? one : (T)0;
    }
}

}  // namespace

MI355_API int32_t mi355_fill_uniform(mi355_ctx *ctx, mi355_stream stream, void *dst, int32_t dtype, uint64_t n,
                                     uint64_t seed, uint64_t tensor, float lo, float hi)
{
    MI355_REQUIRE_CTX(ctx);
    if (n == 0) return MI355_OK;
    if (!dst) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_fill_uniform: dst is NULL");
    const uint64_t key = host_splitmix64(seed ^ (tensor * 0xD6E8FEB86659FD93ull));
    const float scale = hi - lo;
    hipStream_t s = stream_of(ctx, stream);
    const uint32_t grid = grid_for(ctx, n);
    switch (dtype) {
    case MI355_DTYPE_F32:
        hipLaunchKernelGGL(fill_uniform_kernel<MI355_DTYPE_F32>, dim3(grid), dim3(256), 0, s, dst, n, key, lo, scale);
        break;
    case MI355_DTYPE_BF16:
        hipLaunchKernelGGL(fill_uniform_kernel<MI355_DTYPE_BF16>, dim3(grid), dim3(256), 0, s, dst, n, key, lo, scale);
        break;
    case MI355_DTYPE_F16:
        hipLaunchKernelGGL(fill_uniform_kernel<MI355_DTYPE_F16>, dim3(grid), dim3(256), 0, s, dst, n, key, lo, scale);
        break;
    case MI355_DTYPE_U8:            // also what packed fp4 pairs and ue8m0 scales are filled as: n BYTES, floor(uniform[lo, hi))
    case MI355_DTYPE_F4E2M1X2:
    case MI355_DTYPE_UE8M0:
        hipLaunchKernelGGL(fill_uniform_kernel<MI355_DTYPE_U8>, dim3(grid), dim3(256), 0, s, dst, n, key, lo, scale);
        break;
    case MI355_DTYPE_F8E4M3:
        hipLaunchKernelGGL(fill_uniform_kernel<MI355_DTYPE_F8E4M3>, dim3(grid), dim3(256), 0, s, dst, n, key, lo, scale);
        break;
    case MI355_DTYPE_F8E5M2:
        hipLaunchKernelGGL(fill_uniform_kernel<MI355_DTYPE_F8E5M2>, dim3(grid), dim3(256), 0, s, dst, n, key, lo, scale);
        break;
    default:
        return fail(ctx, MI355_E_UNSUPPORTED, "mi355_fill_uniform: unsupported dtype %d", dtype);
    }
    check_launch(ctx, "mi355_fill_uniform");
    return MI355_OK;
}

#define CAST_CASE(S, D)                                                                                      \
    if (src_dtype == (S) && dst_dtype == (D)) {                                                              \
        hipLaunchKernelGGL((cast_kernel<S, D>), dim3(grid), dim3(256), 0, s, src, dst, n);                   \
        launched = true;                                                                                     \
    }

MI355_API int32_t mi355_cast(mi355_ctx *ctx, mi355_stream stream, const void *src, int32_t src_dtype, void *dst,
                             int32_t dst_dtype, uint64_t n)
{
    MI355_REQUIRE_CTX(ctx);
    if (n == 0) return MI355_OK;
    if (!src || !dst) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_cast: NULL pointer");
    hipStream_t s = stream_of(ctx, stream);
    const uint32_t grid = grid_for(ctx, n);
    bool launched = false;
    CAST_CASE(MI355_DTYPE_F32, MI355_DTYPE_BF16)
    CAST_CASE(MI355_DTYPE_F32, MI355_DTYPE_F16)
    CAST_CASE(MI355_DTYPE_BF16, MI355_DTYPE_F32)
    CAST_CASE(MI355_DTYPE_F16, MI355_DTYPE_F32)
    CAST_CASE(MI355_DTYPE_F32, MI355_DTYPE_F32)
    CAST_CASE(MI355_DTYPE_BF16, MI355_DTYPE_F16)
    CAST_CASE(MI355_DTYPE_F16, MI355_DTYPE_BF16)
    CAST_CASE(MI355_DTYPE_F32, MI355_DTYPE_F8E4M3)
    CAST_CASE(MI355_DTYPE_F32, MI355_DTYPE_F8E5M2)
    CAST_CASE(MI355_DTYPE_BF16, MI355_DTYPE_F8E4M3)
    CAST_CASE(MI355_DTYPE_BF16, MI355_DTYPE_F8E5M2)
    CAST_CASE(MI355_DTYPE_F16, MI355_DTYPE_F8E4M3)
    CAST_CASE(MI355_DTYPE_F16, MI355_DTYPE_F8E5M2)
    CAST_CASE(MI355_DTYPE_F8E4M3, MI355_DTYPE_F32)
    CAST_CASE(MI355_DTYPE_F8E5M2, MI355_DTYPE_F32)
    CAST_CASE(MI355_DTYPE_F8E4M3, MI355_DTYPE_BF16)
    CAST_CASE(MI355_DTYPE_F8E5M2, MI355_DTYPE_BF16)
    CAST_CASE(MI355_DTYPE_F8E4M3, MI355_DTYPE_F16)
    CAST_CASE(MI355_DTYPE_F8E5M2, MI355_DTYPE_F16)
    if (!launched) return fail(ctx, MI355_E_UNSUPPORTED, "mi355_cast: unsupported %d -> %d", src_dtype, dst_dtype);
    check_launch(ctx, "mi355_cast");
    return MI355_OK;
}

MI355_API int32_t mi355_fill_identity(mi355_ctx *ctx, mi355_stream stream, void *out, int32_t dtype, uint64_t dim, uint64_t ld)
{
    MI355_REQUIRE_CTX(ctx);
    if (dim == 0) return MI355_OK;
    if (!out) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_fill_identity: out is NULL");
    if (ld < dim) return fail(ctx, MI355_E_UNSUPPORTED_STRIDES, "mi355_fill_identity: row stride %llu < dim %llu", (unsigned long long)ld,
                              (unsigned long long)dim);
    hipStream_t s = stream_of(ctx, stream);
    const uint32_t grid = grid_for(ctx, dim * dim);
    switch (dtype) {
    case MI355_DTYPE_F32: hipLaunchKernelGGL(identity_kernel<uint32_t>, dim3(grid), dim3(256), 0, s, (uint32_t *)out, dim, ld, 0x3F800000u); break;
    case MI355_DTYPE_I32: case MI355_DTYPE_U32:
        hipLaunchKernelGGL(identity_kernel<uint32_t>, dim3(grid), dim3(256), 0, s, (uint32_t *)out, dim, ld, 1u); break;
    case MI355_DTYPE_BF16: hipLaunchKernelGGL(identity_kernel<uint16_t>, dim3(grid), dim3(256), 0, s, (uint16_t *)out, dim, ld, (uint16_t)0x3F80); break;
    case MI355_DTYPE_F16: hipLaunchKernelGGL(identity_kernel<uint16_t>, dim3(grid), dim3(256), 0, s, (uint16_t *)out, dim, ld, (uint16_t)0x3C00); break;
    case MI355_DTYPE_F64:
        hipLaunchKernelGGL(identity_kernel<uint64_t>, dim3(grid), dim3(256), 0, s, (uint64_t *)out, dim, ld, 0x3FF0000000000000ull); break;
    case MI355_DTYPE_I64: case MI355_DTYPE_U64:
        hipLaunchKernelGGL(identity_kernel<uint64_t>, dim3(grid), dim3(256), 0, s, (uint64_t *)out, dim, ld, 1ull); break;
    case MI355_DTYPE_U8: case MI355_DTYPE_I8:
        hipLaunchKernelGGL(identity_kernel<uint8_t>, dim3(grid), dim3(256), 0, s, (uint8_t *)out, dim, ld, (uint8_t)1); break;
    case MI355_DTYPE_F8E4M3: hipLaunchKernelGGL(identity_kernel<uint8_t>, dim3(grid), dim3(256), 0, s, (uint8_t *)out, dim, ld, (uint8_t)0x38); break;
    case MI355_DTYPE_F8E5M2: hipLaunchKernelGGL(identity_kernel<uint8_t>, dim3(grid), dim3(256), 0, s, (uint8_t *)out, dim, ld, (uint8_t)0x3C); break;
    default:
        return fail(ctx, MI355_E_UNSUPPORTED, "mi355_fill_identity: unsupported dtype %d", dtype);
    }
    check_launch(ctx, "mi355_fill_identity");
    return MI355_OK;
}
