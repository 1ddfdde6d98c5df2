// gemm.cpp -- mi355_gemm: descriptor validation and kernel selection.
#include "gemm_common.hpp"

using namespace mi355;

namespace {

int32_t validate(mi355_ctx *ctx, const mi355_gemm_desc *d, const void *a, const void *b, const void *c)
{
    if (!d) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_gemm: descriptor is NULL");
    if (d->m < 0 || d->n < 0 || d->k < 0 || d->batch < 0)
        return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_gemm: negative dimension");
    if (d->dtype_ab != MI355_DTYPE_F32 && d->dtype_ab != MI355_DTYPE_BF16 && d->dtype_ab != MI355_DTYPE_F16)
        return fail(ctx, MI355_E_UNSUPPORTED, "mi355_gemm: unsupported input dtype %d", d->dtype_ab);
    if (d->dtype_c != MI355_DTYPE_F32 && d->dtype_c != d->dtype_ab)
        return fail(ctx, MI355_E_UNSUPPORTED, "mi355_gemm: output dtype %d must be f32 or the input dtype", d->dtype_c);
    if (d->m == 0 || d->n == 0 || d->batch == 0) return MI355_OK;
    if (!c || (d->k > 0 && (!a || !b))) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_gemm: NULL operand");
    const int64_t a_min = d->trans_a ? d->m : d->k, b_min = d->trans_b ? d->k : d->n;
    if (d->lda < a_min || d->ldb < b_min || d->ldc < d->n)
        return fail(ctx, MI355_E_UNSUPPORTED_STRIDES, "mi355_gemm: leading dimension smaller than the row (lda %lld ldb %lld ldc %lld)",
                    (long long)d->lda, (long long)d->ldb, (long long)d->ldc);
    if (d->stride_a < 0 || d->stride_b < 0 || d->stride_c < 0)
        return fail(ctx, MI355_E_UNSUPPORTED_STRIDES, "mi355_gemm: negative batch stride");
    if (d->batch > 1 && d->stride_c < d->m * d->ldc - (d->ldc - d->n))
        return fail(ctx, MI355_E_UNSUPPORTED_STRIDES, "mi355_gemm: output batches overlap");
    return -1;  // proceed
}

int32_t select(const mi355_gemm_desc &d, const void *a, const void *b, const void *c)
{
    if (d.k == 0) return MI355_GEMM_ALGO_GENERIC;  // writes zeros
    if (d.dtype_ab == MI355_DTYPE_F32) {
        // 256x256 tiles (one wave per SIMD) when they give (nearly) every CU a tile; else 128x128
        if (gemm_lp256w4_supports(d, a, b, c) && ((d.m + 255) / 256) * ((d.n + 255) / 256) * d.batch >= 192) return MI355_GEMM_ALGO_LP_256W4;
        if (gemm_f32_mfma_supports(d, a, b, c)) return MI355_GEMM_ALGO_F32_MFMA;
        return MI355_GEMM_ALGO_GENERIC;
    }
    const bool big = gemm_lp256_supports(d, a, b, c);
    const bool big4 = gemm_lp256w4_supports(d, a, b, c);
    const bool mid = gemm_lp128_supports(d, a, b, c);
    if (big) {
        // 256x256 tiles only when they still give every CU work
        const int64_t tiles256 = ((d.m + 255) / 256) * ((d.n + 255) / 256) * d.batch;
        if (tiles256 >= 192 || !mid) return big4 ? MI355_GEMM_ALGO_LP_256W4 : MI355_GEMM_ALGO_LP_256;
    }
    if (mid) return MI355_GEMM_ALGO_LP_128;
    return MI355_GEMM_ALGO_GENERIC;
}

}  // namespace

MI355_API int32_t mi355_gemm_select(mi355_ctx *ctx, const mi355_gemm_desc *desc, int32_t *out_algo)
{
    if (!ctx || !desc || !out_algo) return MI355_E_INVALID_ARGUMENT;
    // alignment-dependent choices are evaluated for 16-byte aligned operands
    static const char aligned_dummy __attribute__((aligned(16))) = 0;
    const mi355_gemm_desc &d = *desc;
    int32_t algo = select(d, &aligned_dummy, &aligned_dummy, &aligned_dummy);
    const bool fix_a = d.trans_a != 0, fix_b = !d.trans_b && d.dtype_ab != MI355_DTYPE_F32;
    if (algo == MI355_GEMM_ALGO_GENERIC && (fix_a || fix_b) && d.k > 0 && d.m * d.n * d.k >= (int64_t)1 << 21) {
        // what mi355_gemm does: re-lay the operand(s) out K-contiguous, then the MFMA kernel (relayout_for_mfma)
        mi355_gemm_desc nd = d;
        const int64_t kpad = (d.k + 7) / 8 * 8;
        if (fix_a) { nd.trans_a = 0; nd.lda = kpad; nd.stride_a = d.stride_a == 0 ? 0 : d.m * kpad; }
        if (fix_b) { nd.trans_b = 1; nd.ldb = kpad; nd.stride_b = d.stride_b == 0 ? 0 : d.n * kpad; }
        algo = select(nd, &aligned_dummy, &aligned_dummy, &aligned_dummy);
    }
    *out_algo = algo;
    return MI355_OK;
}

namespace {

// When AUTO would land on the generic scalar kernel only because of the operand LAYOUT (transposed A, 16-bit
// row-major B), re-lay the operand out into K-contiguous scratch first -- what the reference's launchers do with
// into_contiguous after matrix_batch_layout -- and run the MFMA kernel on it.  Returns MI355_OK and fills `nd`,
// `na`, `nb` when it did; MI355_E_UNSUPPORTED when the layout was not the obstacle (the caller proceeds as before).
int32_t relayout_for_mfma(mi355_ctx *ctx, hipStream_t s, const mi355_gemm_desc &d, const void *a, const void *b, void *c,
                          mi355_gemm_desc &nd, const void *&na, const void *&nb)
{
    const int esz = d.dtype_ab == MI355_DTYPE_F32 ? 4 : 2;
    const bool fix_a = d.trans_a != 0;
    const bool fix_b = !d.trans_b && d.dtype_ab != MI355_DTYPE_F32;       // f32 row-major B has a native kernel
    if (!fix_a && !fix_b) return MI355_E_UNSUPPORTED;
    if (d.k == 0 || d.m * d.n * d.k < (int64_t)1 << 21) return MI355_E_UNSUPPORTED;   // tiny: not worth two launches
    nd = d; na = a; nb = b;
    const int64_t kpad = (d.k + 7) / 8 * 8;                                // 16-byte aligned rows in the scratch
    if (fix_a) {
        const int64_t nba = d.stride_a == 0 ? 1 : d.batch;
        void *p = nullptr;
        if (scratch_get(ctx, s, SCRATCH_RELAYOUT_A, (size_t)(nba * d.m * kpad * esz), &p) != MI355_OK) return MI355_E_UNSUPPORTED;
        launch_transpose(s, a, p, d.k, d.m, d.lda, kpad, nba, d.stride_a, d.m * kpad, esz);    // [K][M] -> [M][K]
        na = p; nd.trans_a = 0; nd.lda = kpad; nd.stride_a = d.stride_a == 0 ? 0 : d.m * kpad;
    }
    if (fix_b) {
        const int64_t nbb = d.stride_b == 0 ? 1 : d.batch;
        void *p = nullptr;
        if (scratch_get(ctx, s, SCRATCH_RELAYOUT_B, (size_t)(nbb * d.n * kpad * esz), &p) != MI355_OK) return MI355_E_UNSUPPORTED;
        launch_transpose(s, b, p, d.k, d.n, d.ldb, kpad, nbb, d.stride_b, d.n * kpad, esz);    // [K][N] -> [N][K]
        nb = p; nd.trans_b = 1; nd.ldb = kpad; nd.stride_b = d.stride_b == 0 ? 0 : d.n * kpad;
    }
    check_launch(ctx, "mi355_gemm(operand re-layout)");
    if (select(nd, na, nb, c) == MI355_GEMM_ALGO_GENERIC) return MI355_E_UNSUPPORTED;   // e.g. ragged K: nothing gained
    return MI355_OK;
}

}  // namespace

MI355_API int32_t mi355_gemm(mi355_ctx *ctx, mi355_stream stream, const mi355_gemm_desc *desc, const void *a,
                             const void *b, void *c)
{
    MI355_REQUIRE_CTX(ctx);
    int32_t rc = validate(ctx, desc, a, b, c);
    if (rc != -1) return rc;
    hipStream_t s = stream_of(ctx, stream);
    mi355_gemm_desc d = *desc;
    int32_t algo = d.algo == MI355_GEMM_ALGO_AUTO ? select(d, a, b, c) : d.algo;
    if (d.algo == MI355_GEMM_ALGO_AUTO && algo == MI355_GEMM_ALGO_GENERIC) {
        mi355_gemm_desc nd;
        const void *na, *nb;
        if (relayout_for_mfma(ctx, s, d, a, b, c, nd, na, nb) == MI355_OK) {
            d = nd; a = na; b = nb;
            algo = select(d, a, b, c);
        }
    }
    switch (algo) {
    case MI355_GEMM_ALGO_GENERIC: return launch_gemm_generic(ctx, s, d, a, b, c);
    case MI355_GEMM_ALGO_F32_MFMA: return launch_gemm_f32_mfma(ctx, s, d, a, b, c);
    case MI355_GEMM_ALGO_LP_128: return launch_gemm_lp128(ctx, s, d, a, b, c);
    case MI355_GEMM_ALGO_LP_256: return launch_gemm_lp256(ctx, s, d, a, b, c);
    case MI355_GEMM_ALGO_LP_256W4: return launch_gemm_lp256w4(ctx, s, d, a, b, c);
    case MI355_GEMM_ALGO_LP_256P: return launch_gemm_lp256p(ctx, s, d, a, b, c);
    default: return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_gemm: unknown algo %d", algo);
    }
}
