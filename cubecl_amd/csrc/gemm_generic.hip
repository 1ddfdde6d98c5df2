// gemm_generic.hip -- bounds-checked LDS-tiled FMA GEMM for every shape / layout / dtype the
// MFMA kernels do not take (ragged K, transposed A, 16-bit row-major B, unaligned strides).
// Correctness path, not a roofline path: 64x64x16 tile, 4x4 outputs per thread, f32 FMA chain in
// k order (the order of runtime_tests/cmma.rs:695-722).
#include "gemm_common.hpp"
#include "fp8.hpp"

#include <hip/hip_fp16.h>

using namespace mi355;

namespace {

template <int DT>
__device__ __forceinline__ float load_as_f32(const void *p, int64_t idx)
{
    if (DT == MI355_DTYPE_F32) return static_cast<const float *>(p)[idx];
    if (DT == MI355_DTYPE_F8E4M3) return e4m3_to_f32(static_cast<const uint8_t *>(p)[idx]);
    if (DT == MI355_DTYPE_F8E5M2) return e5m2_to_f32(static_cast<const uint8_t *>(p)[idx]);
    const uint16_t raw = static_cast<const uint16_t *>(p)[idx];
    if (DT == MI355_DTYPE_BF16) return __uint_as_float((uint32_t)raw << 16);
    return __half2float(__ushort_as_half(raw));
}

template <int DT_AB, int DT_C>
__global__ void __launch_bounds__(256)
gemm_generic_kernel(const void *__restrict__ A, const void *__restrict__ B, void *__restrict__ C, int64_t M, int64_t N,
                    int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int64_t sa, int64_t sb, int64_t sc, int trans_a,
                    int trans_b, uint32_t tiles_n)
{
    constexpr int BM = 64, BN = 64, BK = 16;
    __shared__ float As[BK][BM + 1];
    __shared__ float Bs[BK][BN + 1];
    const int tid = threadIdx.x;
    const int64_t batch = blockIdx.y;
    const int64_t m0 = (int64_t)(blockIdx.x / tiles_n) * BM;
    const int64_t n0 = (int64_t)(blockIdx.x % tiles_n) * BN;
    const int64_t oa = batch * sa, ob = batch * sb, oc = batch * sc;
    const int tx = tid & 15, ty = tid >> 4;  // thread -> 4 rows (ty*4..) x 4 cols (tx*4..)

    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

    for (int64_t k0 = 0; k0 < K; k0 += BK) {
        // 64x16 elements per operand, 256 threads -> 4 each
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int lin = tid + e * 256;
            {
                const int mm = trans_a ? (lin & 63) : (lin >> 4);
                const int kk = trans_a ? (lin >> 6) : (lin & 15);
                const int64_t m = m0 + mm, k = k0 + kk;
                float v = 0.f;
                if (m < M && k < K) v = load_as_f32<DT_AB>(A, oa + (trans_a ? k * lda + m : m * lda + k));
                As[kk][mm] = v;
            }
            {
                const int nn = trans_b ? (lin >> 4) : (lin & 63);
                const int kk = trans_b ? (lin & 15) : (lin >> 6);
                const int64_t n = n0 + nn, k = k0 + kk;
                float v = 0.f;
                if (n < N && k < K) v = load_as_f32<DT_AB>(B, ob + (trans_b ? n * ldb + k : k * ldb + n));
                Bs[kk][nn] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < BK; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[kk][ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Bs[kk][tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __fmaf_rn(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t m = m0 + ty * 4 + i;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t n = n0 + tx * 4 + j;
            if (n >= N) continue;
            const int64_t idx = oc + m * ldc + n;
            if (DT_C == MI355_DTYPE_F32) static_cast<float *>(C)[idx] = acc[i][j];
            else static_cast<uint16_t *>(C)[idx] = f32_to_lp<DT_C>(acc[i][j]);
        }
    }
}

template <int DT_AB, int DT_C>
void launch(hipStream_t s, const mi355_gemm_desc &d, const void *a, const void *b, void *c)
{
    const uint32_t tiles_m = (uint32_t)((d.m + 63) / 64), tiles_n = (uint32_t)((d.n + 63) / 64);
    hipLaunchKernelGGL((gemm_generic_kernel<DT_AB, DT_C>), dim3(tiles_m * tiles_n, (uint32_t)d.batch), dim3(256), 0, s, a,
                       b, c, d.m, d.n, d.k, d.lda, d.ldb, d.ldc, d.stride_a, d.stride_b, d.stride_c, d.trans_a,
                       d.trans_b, tiles_n);
}

}  // namespace

namespace mi355 {

int32_t launch_gemm_generic(mi355_ctx *ctx, hipStream_t s, const mi355_gemm_desc &d, const void *a, const void *b,
                            void *c)
{
    if (d.batch > 65535) return fail(ctx, MI355_E_UNSUPPORTED, "generic GEMM: batch %lld > 65535", (long long)d.batch);
    const int ab = d.dtype_ab, cd = d.dtype_c;
    if (ab == MI355_DTYPE_F32 && cd == MI355_DTYPE_F32) launch<MI355_DTYPE_F32, MI355_DTYPE_F32>(s, d, a, b, c);
    else if (ab == MI355_DTYPE_BF16 && cd == MI355_DTYPE_F32) launch<MI355_DTYPE_BF16, MI355_DTYPE_F32>(s, d, a, b, c);
    else if (ab == MI355_DTYPE_BF16 && cd == MI355_DTYPE_BF16) launch<MI355_DTYPE_BF16, MI355_DTYPE_BF16>(s, d, a, b, c);
    else if (ab == MI355_DTYPE_F16 && cd == MI355_DTYPE_F32) launch<MI355_DTYPE_F16, MI355_DTYPE_F32>(s, d, a, b, c);
    else if (ab == MI355_DTYPE_F16 && cd == MI355_DTYPE_F16) launch<MI355_DTYPE_F16, MI355_DTYPE_F16>(s, d, a, b, c);
    else if (ab == MI355_DTYPE_F8E4M3 && cd == MI355_DTYPE_F32) launch<MI355_DTYPE_F8E4M3, MI355_DTYPE_F32>(s, d, a, b, c);
    else if (ab == MI355_DTYPE_F8E4M3 && cd == MI355_DTYPE_BF16) launch<MI355_DTYPE_F8E4M3, MI355_DTYPE_BF16>(s, d, a, b, c);
    else if (ab == MI355_DTYPE_F8E4M3 && cd == MI355_DTYPE_F16) launch<MI355_DTYPE_F8E4M3, MI355_DTYPE_F16>(s, d, a, b, c);
    else if (ab == MI355_DTYPE_F8E5M2 && cd == MI355_DTYPE_F32) launch<MI355_DTYPE_F8E5M2, MI355_DTYPE_F32>(s, d, a, b, c);
    else if (ab == MI355_DTYPE_F8E5M2 && cd == MI355_DTYPE_BF16) launch<MI355_DTYPE_F8E5M2, MI355_DTYPE_BF16>(s, d, a, b, c);
    else if (ab == MI355_DTYPE_F8E5M2 && cd == MI355_DTYPE_F16) launch<MI355_DTYPE_F8E5M2, MI355_DTYPE_F16>(s, d, a, b, c);
    else return fail(ctx, MI355_E_UNSUPPORTED, "generic GEMM: unsupported dtypes ab=%d c=%d", ab, cd);
    check_launch(ctx, "mi355_gemm(generic)");
    return MI355_OK;
}

}  // namespace mi355
