// gemm_scaled.hip -- mi355_gemm_scaled: block-scaled (MX) matmul, C = (A .* SA) (B .* SB)^T.
//
// What it replaces: `MmaDefinition::execute_scaled` (crates/cubecl-core/src/frontend/cmma.rs:795-840) as a cubek-style
// matmul launcher would tile it; semantics pinned by test_cmma_scaled / test_cmma_scaled_fp4
// (crates/cubecl-core/src/runtime_tests/cmma.rs:1476-1704): one ue8m0 scale per `block` consecutive k-values of a row of
// A ([M][K]) and of B (stored [N][K]).
//
// Two paths:
//   * block == 32, K-tile-multiple K, aligned rows: the 256x256 MFMA kernel (gemm_lp256w4.hip, MX form) on
//     v_mfma_scale_f32_32x32x64_f8f6f4 -- roofline MFMA fp8 ~5 PFLOP/s / fp4 ~10 PFLOP/s dense.  The hardware wants each
//     lane's scale byte in a register; fetched from the caller's [rows][K/32] layout that would be a 64-lane gather over
//     32 cache lines per instruction, 8 instructions per K-tile and wave -- as much L1 work as the operand stream itself.
//     So the scales are first re-arranged into ST[K-tile][row, padded to the tile grid][blocks per K-tile row] in
//     library-owned per-stream scratch (rearrange_scales_kernel: reads and writes M x K/32 bytes once, ~2 MB for 8192^2,
//     i.e. microseconds), after which a lane's share is one coalesced 2- / 4-byte load per 32-row block and K-tile.
//   * everything else: scaled_generic_kernel, one thread per output, the reference loop literally (f32, left to
//     right: lhs * lhs_scale * rhs * rhs_scale).  Correctness net, not a roofline path.
#include <algorithm>
#include <type_traits>

#include "fp8.hpp"
#include "gemm_common.hpp"

using namespace mi355;

namespace {

__device__ __forceinline__ float ue8m0_to_f32(uint8_t b)
{
    if (b == 0xFFu) return __uint_as_float(0x7FC00000u);
    if (b == 0u) return __uint_as_float(0x00400000u);         // 2^-127 (a subnormal f32)
    return __uint_as_float((uint32_t)b << 23);
}

__device__ __forceinline__ float e2m1_to_f32(uint32_t nibble)
{
    // {0, 0.5, 1, 1.5, 2, 3, 4, 6}: code c >= 2 is (1 + (c&1)/2) * 2^((c>>1) - 1), codes 0 / 1 are 0 / 0.5
    const uint32_t c = nibble & 7u;
    const float v = c < 2u ? 0.5f * (float)c : __uint_as_float(((126u + (c >> 1)) << 23) | ((c & 1u) << 22));
    return (nibble & 8u) ? -v : v;
}

template <int DT>
__device__ __forceinline__ float load_mx(const void *p, int64_t idx)
{
    if (DT == MI355_DTYPE_F8E4M3) return e4m3_to_f32(static_cast<const uint8_t *>(p)[idx]);
    if (DT == MI355_DTYPE_F8E5M2) return e5m2_to_f32(static_cast<const uint8_t *>(p)[idx]);
    return e2m1_to_f32((uint32_t)(static_cast<const uint8_t *>(p)[idx >> 1] >> ((idx & 1) * 4)));
}

template <int DTA, int DTB, int DT_C>
__global__ void __launch_bounds__(256)
scaled_generic_kernel(const void *__restrict__ A, const uint8_t *__restrict__ SA, const void *__restrict__ B,
                      const uint8_t *__restrict__ SB, void *__restrict__ C, mi355_gemm_scaled_desc d)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= d.m * d.n) return;
    const int64_t bt = blockIdx.y, m = idx / d.n, n = idx % d.n;
    const int64_t oa = bt * d.stride_a + m * d.lda, ob = bt * d.stride_b + n * d.ldb;
    const uint8_t *sa = SA + bt * d.stride_sa + m * d.ld_sa, *sb = SB + bt * d.stride_sb + n * d.ld_sb;
    float sum = 0.f;
    for (int64_t k0 = 0; k0 < d.k; k0 += d.block) {
        const float ls = ue8m0_to_f32(sa[k0 / d.block]), rs = ue8m0_to_f32(sb[k0 / d.block]);
        const int64_t kend = min(k0 + (int64_t)d.block, d.k);
        for (int64_t k = k0; k < kend; ++k) {
            float p = __fmul_rn(load_mx<DTA>(A, oa + k), ls);             // separate roundings, as the reference loop
            p = __fmul_rn(p, load_mx<DTB>(B, ob + k));
            p = __fmul_rn(p, rs);
            sum = __fadd_rn(sum, p);
        }
    }
    const int64_t o = bt * d.stride_c + m * d.ldc + n;
    if (DT_C == MI355_DTYPE_F32) static_cast<float *>(C)[o] = sum;
    else static_cast<uint16_t *>(C)[o] = f32_to_lp<DT_C>(sum);
}

// ST[t][r][0..NB) = S[r][t*NB .. t*NB+NB) for r < rows, the neutral scale 2^0 (0x7F) for the padding rows.
template <int NB>
__global__ void __launch_bounds__(256)
rearrange_scales_kernel(const uint8_t *__restrict__ S, uint8_t *__restrict__ ST, int64_t rows, int64_t rows_pad, int64_t ktiles,
                        int64_t ld_s, int64_t stride_s, int64_t stride_st)
{
    typedef typename std::conditional<NB == 4, uint32_t, uint64_t>::type unit;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows_pad * ktiles) return;
    const int64_t t = idx / rows_pad, r = idx % rows_pad;                 // consecutive threads -> consecutive rows: coalesced writes
    const uint8_t *src = S + (int64_t)blockIdx.y * stride_s + r * ld_s + t * NB;
    unit v;
    if (r >= rows) v = (unit)0x7F7F7F7F7F7F7F7Full;
    else {
        v = 0;
#pragma unroll
        for (int b = 0; b < NB; ++b) v |= (unit)src[b] << (8 * b);
    }
    reinterpret_cast<unit *>(ST + (int64_t)blockIdx.y * stride_st)[idx] = v;
}

int32_t validate(mi355_ctx *ctx, const mi355_gemm_scaled_desc *d, const void *a, const void *sa, const void *b, const void *sb,
                 const void *c)
{
    if (!d) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_gemm_scaled: descriptor is NULL");
    if (d->m < 0 || d->n < 0 || d->k < 0 || d->batch < 0) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_gemm_scaled: negative dimension");
    const bool f4a = d->dtype_a == MI355_DTYPE_F4E2M1X2, f4b = d->dtype_b == MI355_DTYPE_F4E2M1X2;
    if (!(f4a || is_fp8(d->dtype_a)) || !(f4b || is_fp8(d->dtype_b)) || f4a != f4b)
        return fail(ctx, MI355_E_UNSUPPORTED, "mi355_gemm_scaled: operand dtypes %d x %d (fp8 x fp8, mixed formats allowed, or fp4 x fp4)",
                    d->dtype_a, d->dtype_b);
    if (d->dtype_c != MI355_DTYPE_F32 && d->dtype_c != MI355_DTYPE_BF16 && d->dtype_c != MI355_DTYPE_F16)
        return fail(ctx, MI355_E_UNSUPPORTED, "mi355_gemm_scaled: output dtype %d (f32, bf16 or f16)", d->dtype_c);
    if (d->block < 1 || (d->k % d->block) != 0)
        return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_gemm_scaled: K %lld is not a multiple of the scale block %d", (long long)d->k, d->block);
    if (f4a && ((d->k | d->lda | d->ldb | d->stride_a | d->stride_b) & 1))
        return fail(ctx, MI355_E_UNSUPPORTED_STRIDES, "mi355_gemm_scaled: packed fp4 needs even K, leading dimensions and batch strides");
    if (d->m == 0 || d->n == 0 || d->batch == 0) return MI355_OK;
    if (!c || (d->k > 0 && (!a || !b || !sa || !sb))) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_gemm_scaled: NULL operand");
    if (d->lda < d->k || d->ldb < d->k || d->ldc < d->n || d->ld_sa < d->k / d->block || d->ld_sb < d->k / d->block)
        return fail(ctx, MI355_E_UNSUPPORTED_STRIDES, "mi355_gemm_scaled: leading dimension smaller than the row");
    if (d->stride_a < 0 || d->stride_b < 0 || d->stride_c < 0 || d->stride_sa < 0 || d->stride_sb < 0)
        return fail(ctx, MI355_E_UNSUPPORTED_STRIDES, "mi355_gemm_scaled: negative batch stride");
    if (d->batch > 65535) return fail(ctx, MI355_E_UNSUPPORTED, "mi355_gemm_scaled: batch %lld > 65535", (long long)d->batch);
    return -1;
}

int32_t select(const mi355_gemm_scaled_desc &d, const void *a, const void *b, const void *c)
{
    if (d.k == 0) return MI355_GEMM_ALGO_GENERIC;
    // the scale bytes are read one row-unit at a time by the re-arrangement: any ld_s works
    if (gemm_lp256w4_mx_supports(d, a, b, c) && d.m * d.n * d.k >= ((int64_t)1 << 21)) return MI355_GEMM_ALGO_LP_256W4;
    return MI355_GEMM_ALGO_GENERIC;
}

template <int DTA, int DTB>
void launch_generic_c(hipStream_t s, const mi355_gemm_scaled_desc &d, const void *a, const void *sa, const void *b, const void *sb, void *c)
{
    const dim3 grid((uint32_t)((d.m * d.n + 255) / 256), (uint32_t)d.batch);
    const uint8_t *psa = static_cast<const uint8_t *>(sa), *psb = static_cast<const uint8_t *>(sb);
    if (d.dtype_c == MI355_DTYPE_F32)
        hipLaunchKernelGGL((scaled_generic_kernel<DTA, DTB, MI355_DTYPE_F32>), grid, dim3(256), 0, s, a, psa, b, psb, c, d);
    else if (d.dtype_c == MI355_DTYPE_BF16)
        hipLaunchKernelGGL((scaled_generic_kernel<DTA, DTB, MI355_DTYPE_BF16>), grid, dim3(256), 0, s, a, psa, b, psb, c, d);
    else
        hipLaunchKernelGGL((scaled_generic_kernel<DTA, DTB, MI355_DTYPE_F16>), grid, dim3(256), 0, s, a, psa, b, psb, c, d);
}

int32_t launch_generic(mi355_ctx *ctx, hipStream_t s, const mi355_gemm_scaled_desc &d, const void *a, const void *sa, const void *b,
                       const void *sb, void *c)
{
    if (d.m * d.n > (int64_t)0x7FFFFFFF * 256) return fail(ctx, MI355_E_UNSUPPORTED, "generic block-scaled GEMM: output too large");
    constexpr int E4 = MI355_DTYPE_F8E4M3, E5 = MI355_DTYPE_F8E5M2, F4 = MI355_DTYPE_F4E2M1X2;
    if (d.dtype_a == F4) launch_generic_c<F4, F4>(s, d, a, sa, b, sb, c);
    else if (d.dtype_a == E4 && d.dtype_b == E4) launch_generic_c<E4, E4>(s, d, a, sa, b, sb, c);
    else if (d.dtype_a == E5 && d.dtype_b == E5) launch_generic_c<E5, E5>(s, d, a, sa, b, sb, c);
    else if (d.dtype_a == E4) launch_generic_c<E4, E5>(s, d, a, sa, b, sb, c);
    else launch_generic_c<E5, E4>(s, d, a, sa, b, sb, c);
    check_launch(ctx, "mi355_gemm_scaled(generic)");
    return MI355_OK;
}

// Re-arranges one operand's scales into scratch; returns the scratch pointer and its batch stride (bytes).
int32_t rearrange(mi355_ctx *ctx, hipStream_t s, int kind, const void *scales, int64_t rows, int64_t k, int nb, int64_t ld_s,
                  int64_t stride_s, int64_t batch, const void **out, int64_t *out_stride)
{
    const int64_t rows_pad = (rows + 255) / 256 * 256, ktiles = k / (32 * nb);
    const int64_t nbatch = stride_s == 0 ? 1 : batch;                      // a broadcast operand is re-arranged once
    const int64_t per_batch = rows_pad * ktiles * nb;
    void *q = nullptr;
    const int32_t rc = scratch_get(ctx, s, kind, (size_t)(per_batch * nbatch), &q);
    if (rc != MI355_OK) return fail(ctx, rc, "mi355_gemm_scaled: no scratch for the re-arranged scales");
    const dim3 grid((uint32_t)((rows_pad * ktiles + 255) / 256), (uint32_t)nbatch);
    const uint8_t *src = static_cast<const uint8_t *>(scales);
    if (nb == 4)
        hipLaunchKernelGGL(rearrange_scales_kernel<4>, grid, dim3(256), 0, s, src, static_cast<uint8_t *>(q), rows, rows_pad, ktiles, ld_s,
                           stride_s, per_batch);
    else
        hipLaunchKernelGGL(rearrange_scales_kernel<8>, grid, dim3(256), 0, s, src, static_cast<uint8_t *>(q), rows, rows_pad, ktiles, ld_s,
                           stride_s, per_batch);
    *out = q;
    *out_stride = stride_s == 0 ? 0 : per_batch;
    return MI355_OK;
}

}  // namespace

MI355_API int32_t mi355_gemm_scaled(mi355_ctx *ctx, mi355_stream stream, const mi355_gemm_scaled_desc *desc, const void *a,
                                    const void *a_scales, const void *b, const void *b_scales, void *c)
{
    MI355_REQUIRE_CTX(ctx);
    int32_t rc = validate(ctx, desc, a, a_scales, b, b_scales, c);
    if (rc != -1) return rc;
    hipStream_t s = stream_of(ctx, stream);
    const mi355_gemm_scaled_desc &d = *desc;
    const int32_t algo = d.algo == MI355_GEMM_ALGO_AUTO ? select(d, a, b, c) : d.algo;
    if (algo == MI355_GEMM_ALGO_GENERIC) return launch_generic(ctx, s, d, a, a_scales, b, b_scales, c);
    if (algo != MI355_GEMM_ALGO_LP_256W4) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_gemm_scaled: unknown algo %d", algo);
    if (!gemm_lp256w4_mx_supports(d, a, b, c))
        return fail(ctx, MI355_E_UNSUPPORTED, "mi355_gemm_scaled: the MFMA kernel does not take this shape / layout / block size");
    const int nb = d.dtype_a == MI355_DTYPE_F4E2M1X2 ? 8 : 4;
    const void *sa_t = nullptr, *sb_t = nullptr;
    int64_t st_a = 0, st_b = 0;
    rc = rearrange(ctx, s, SCRATCH_SCALE_A, a_scales, d.m, d.k, nb, d.ld_sa, d.stride_sa, d.batch, &sa_t, &st_a);
    if (rc != MI355_OK) return rc;
    rc = rearrange(ctx, s, SCRATCH_SCALE_B, b_scales, d.n, d.k, nb, d.ld_sb, d.stride_sb, d.batch, &sb_t, &st_b);
    if (rc != MI355_OK) return rc;
    check_launch(ctx, "mi355_gemm_scaled(scale re-arrangement)");
    return launch_gemm_lp256w4_mx(ctx, s, d, a, sa_t, st_a, b, sb_t, st_b, c);
}

MI355_API int32_t mi355_gemm_scaled_select(mi355_ctx *ctx, const mi355_gemm_scaled_desc *desc, int32_t *out_algo)
{
    if (!ctx || !desc || !out_algo) return MI355_E_INVALID_ARGUMENT;
    static const char aligned_dummy __attribute__((aligned(16))) = 0;
    *out_algo = select(*desc, &aligned_dummy, &aligned_dummy, &aligned_dummy);
    return MI355_OK;
}
