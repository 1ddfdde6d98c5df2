// gemm.cpp -- mi355_gemm: descriptor validation and kernel selection.
#include "gemm_common.hpp"

using namespace mi355;

namespace {

int32_t validate(mi355_ctx *ctx, const mi355_gemm_desc *d, const void *a, const void *b, const void *c)
{
    if (!d) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_gemm: descriptor is NULL");
    if (d->m < 0 || d->n < 0 || d->k < 0 || d->batch < 0)
        return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_gemm: negative dimension");
    const bool f8 = is_fp8(d->dtype_ab);
    if (d->dtype_ab != MI355_DTYPE_F32 && d->dtype_ab != MI355_DTYPE_BF16 && d->dtype_ab != MI355_DTYPE_F16 && !f8)
        return fail(ctx, MI355_E_UNSUPPORTED, "mi355_gemm: unsupported input dtype %d", d->dtype_ab);
    if (f8) {
        if (d->dtype_c != MI355_DTYPE_F32 && d->dtype_c != MI355_DTYPE_BF16 && d->dtype_c != MI355_DTYPE_F16)
            return fail(ctx, MI355_E_UNSUPPORTED, "mi355_gemm: fp8 inputs write f32, bf16 or f16 (got %d)", d->dtype_c);
    } else if (d->dtype_c != MI355_DTYPE_F32 && d->dtype_c != d->dtype_ab)
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

int32_t select(const mi355_gemm_desc &d, const void *a, const void *b, const void *c, bool strip_kernel = true)
{
    if (d.k == 0) return MI355_GEMM_ALGO_GENERIC;  // writes zeros
    if (is_fp8(d.dtype_ab)) {   // 256x256 tiles from 129 tiles up (as for bf16), the 128x128 kernel below that
        const bool big4 = gemm_lp256w4_supports(d, a, b, c), mid = gemm_lp128_supports(d, a, b, c);
        const int64_t tiles256 = ((d.m + 255) / 256) * ((d.n + 255) / 256) * d.batch;
        if (big4 && (tiles256 > 128 || !mid) && d.m * d.n * d.k >= ((int64_t)1 << 21)) return MI355_GEMM_ALGO_LP_256W4;
        if (mid) return MI355_GEMM_ALGO_LP_128;
        return MI355_GEMM_ALGO_GENERIC;
    }
    if (d.dtype_ab == MI355_DTYPE_F32) {
        // at most 8 rows (or columns): the row-streaming FMA kernel (gemm_skinny.hip, round 4; profiles/r04_f32_audit.txt, us, against
        // the 128x128 tile kernel): 1 / 2 / 4 / 8 x 8192 x 8192 43.2 / 47.6 / 53.6 / 73.9 against 150-161, 1 x 4096 x 4096 12.9 / 42.2,
        // 8 x 2048 x 8192 25.3 / 44.0; with 16 rows its 64 FMAs per 16 bytes lose (156 / 152, 16 x 4096 x 4096 55.8 / 43.4)
        if (std::min(d.m, d.n) <= 8 && gemm_skinny_supports(d, a, b, c)) return MI355_GEMM_ALGO_SKINNY;
        // ... and against a ROW-MAJOR f32 weight the strip kernel's f32 form (gemm_nnrows.hip on v_mfma_f32_4x4x1: no transposition at
        // all), up to 16 rows (profiles/r04_f32_audit.txt, us, against the 128x128 tile kernel): 1 / 8 / 16 x 8192 x 8192 45.8 / 49.0 / 57.7
        // against 151-158, 16 x 4096 x 4096 19.8 / 46.3, 4 x 28672 x 4096 75 / 294, 1 x 2048 x 1024 10.6 / 23.1
        // (`strip_kernel` false: the capture-window fallback of mi355_gemm -- the strip kernel's scratch / tickets cannot be created
        // while a stream is capturing -- must reach the tile kernels here as it does for 16-bit operands)
        if (strip_kernel && !d.trans_a && !d.trans_b && d.m <= 16 && d.batch == 1 && d.n * d.k >= ((int64_t)1 << 16) && gemm_nnrows_supports(d, a, b, c))
            return MI355_GEMM_ALGO_NNROWS;
        // 256x256 tiles (one wave per SIMD) when they give (nearly) every CU a tile; else 128x128
        if (gemm_lp256w4_supports(d, a, b, c) && ((d.m + 255) / 256) * ((d.n + 255) / 256) * d.batch >= 192) return MI355_GEMM_ALGO_LP_256W4;
        if (gemm_f32_mfma_supports(d, a, b, c)) return MI355_GEMM_ALGO_F32_MFMA;
        return MI355_GEMM_ALGO_GENERIC;
    }
    // A stored [K][M] with a row-major B (lhs^T . grad_out, the weight-gradient product of a training step): the 128x128 kernel
    // stages both natively (gemm_lp128.hip ATN) -- taken exactly when that kernel would be chosen for the same shape with a
    // K-contiguous A, and so does the 256x256 kernel; where another kernel would win (the streaming ones, the 256 x 128 tile, the
    // persistent forms' ground is taken by the one-tile kernel), A is transposed into scratch first (GENERIC = "re-lay out, then
    // select again") and B stays as it is.
    if (d.trans_a) {
        if (d.trans_b) return MI355_GEMM_ALGO_GENERIC;
        static const char aligned_dummy __attribute__((aligned(16))) = 0;
        mi355_gemm_desc e = d;
        e.trans_a = 0; e.lda = d.k; e.stride_a = d.stride_a == 0 ? 0 : d.m * d.k;
        const int32_t twin = select(e, &aligned_dummy, b, c);             // what the same shape takes with a K-contiguous A
        if (twin == MI355_GEMM_ALGO_LP_128 && gemm_lp128_supports(d, a, b, c)) return MI355_GEMM_ALGO_LP_128;
        // ... and the 256x256 kernel stages it the same way (gemm_lp256w4.hip ATN; its persistent forms do not: select_auto leaves
        // a transposed lhs on the one-tile-per-workgroup kernel)
        if (twin == MI355_GEMM_ALGO_LP_256W4 && gemm_lp256w4_supports(d, a, b, c)) return MI355_GEMM_ALGO_LP_256W4;
        return MI355_GEMM_ALGO_GENERIC;
    }
    // (the 8-wave 256x256 kernel, gemm_lp256.hip, was retired in round 5: since the 4-wave kernel takes unaligned C rows no AUTO path
    // reached it; MI355_GEMM_ALGO_LP_256 stays as an alias of the 4-wave kernel for callers that name it)
    const bool big4 = gemm_lp256w4_supports(d, a, b, c);
    const bool mid = gemm_lp128_supports(d, a, b, c);
    // Row-major B (the layout TensorHandle::new_contiguous gives a rhs): staged natively by the tile kernels that build the
    // transposing-read image (gemm_lp256w4.hip "BNN"); the streaming / dot-product kernels for few rows only exist for
    // K-contiguous operands, so a launch they would win goes through the re-layout pass (GENERIC here = "re-lay out, then
    // select again" in mi355_gemm) -- a 16 x 8192 x 8192 product costs 22 us there + 50 us of transposition against ~140 us
    // on 32 tiles of 256x256.
    // Round 3, late: that held while B was the SMALL operand.  With few ROWS of A (the decode case: x [M][K] times a row-major
    // weight [K][N]) B is the streamed operand and the transposition pass reads and writes all of it before the product reads it
    // again -- cold operands, interleaved (profiles/r03_nn_few_rows.txt), re-layout + streaming kernel against the 128x128 kernel's
    // native NN form (split-K): 1 x 8192 x 8192 69.6 -> 35.0 us, 16 x 8192 x 8192 64.7 -> 35.9, 64 x 8192 x 8192 67.0 -> 41.7,
    // 16 x 28672 x 8192 270 -> 78.7 (the [N][K] form on its streaming kernel: 76.1), 64 x 28672 x 8192 290 -> 85.2 ([N][K]: 93.1),
    // 8 x 57344 x 4096 276 -> 81.3, 32 x 4096 x 4096 24.4 -> 16.7.  Few COLUMNS (B small: at most 64 x K) keep the re-layout.
    if (!d.trans_b) {
        // Round 4: gemm_nnrows.hip (wide row strips, transposition in registers, no split-K launch pair) where the 128x128 kernel
        // below is weakest -- few column tiles, hence many K slices and a fold launch (N <= 16384), or more than two rounds of
        // tiles (N > 65536) -- and the weight is worth streaming.  Cold operands, us, nnrows against lp128 (profiles/r04_nnrows_ab.txt):
        // 1 / 4 / 8 x 8192 x 8192 25.6 / 26.5 / 27.8 against 30.9 (16 rows: a tie), 4 x 14336 x 4096 23.2 / 25.7, 4 x 4096 x 14336
        // 23.9 / 25.9, 1 / 16 x 128256 x 4096 151 / 191 and 176 / 195; not taken: 4 x 32000 x 4096 44.6 / 40.8 (250 tiles, no
        // split), 4 x 4096 x 4096 13.8 / 11.7, 16 x 28672 x 8192 85 / 76.
        // ... or a second round of tiles it fills by less than three quarters (1 x 40568 x 3072: 317 tiles, 54.7 against 48.2)
        const int64_t t128 = (d.n + 127) / 128;
        if (strip_kernel && d.batch == 1 && d.n * d.k >= ((int64_t)1 << 25) &&
            ((d.m <= 8 && (d.n <= 16384 || (t128 > 256 && t128 < 448))) || (d.m <= 16 && d.n > 65536) ||
             // 9-16 rows since they run on v_mfma_f32_16x16x16 (256-byte strips): 16 x 8192 x 8192 28.7 against 31.3, 16 x 14336 x 4096
             // 25.4 / 26.7, 16 x 16384 x 4096 28.5 / 30.9; not below N = 8192 (16 x 6144 x 6144 21.0 / 19.4, 16 x 4096 x 14336 27.0 / 25.9)
             (d.m <= 16 && d.n >= 8192 && d.n <= 16384)) &&
            gemm_nnrows_supports(d, a, b, c))
            return MI355_GEMM_ALGO_NNROWS;
        const int64_t tiles256 = ((d.m + 255) / 256) * ((d.n + 255) / 256) * d.batch;
        const bool native = (std::min(d.m, d.n) > 128 && big4 && tiles256 > 128) || (std::min(d.m, d.n) > 64 && mid);
        if (!native) {
            if (mid && d.m <= d.n) return MI355_GEMM_ALGO_LP_128;
            // Few COLUMNS (B is the small operand): the re-layout pass (one more launch, ~5 us) pays for itself only where the
            // streaming kernels then win by more than that.  Round 4 (profiles/r04_few_columns_nn_ab.txt, us, re-layout + streaming
            // kernel / 128x128 kernel on the row-major B as it is): where the K-contiguous twin would take the 128x128 kernel
            // anyway the pass was pure loss (32768 x 64 x 512 13.1 / 10.2, 4096 x 64 x 8192 28.6 / 25.7); up to 16 columns the
            // tile kernel wins or ties throughout (16384 x 16 x 128 11.3 / 6.1, 32768 x 16 x 512 18.5 / 9.7, 8192 x 16 x 2048
            // 17.2 / 13.4; at K = 8192 a tie: 36.1 / 37.8, the pass stays); 64 columns keep it (8192 x 64 x 512 9.5 / 13.7, x 2048 14.2 / 18.1).
            if (mid) {
                static const char aligned_dummy __attribute__((aligned(16))) = 0;
                mi355_gemm_desc e = d;                  // the K-contiguous twin: B re-laid out [N][K]
                e.trans_b = 1; e.ldb = d.k; e.stride_b = d.stride_b ? d.n * d.k : 0;
                if ((d.n <= 16 && d.k <= 4096) || select(e, a, &aligned_dummy, c, false) == MI355_GEMM_ALGO_LP_128) return MI355_GEMM_ALGO_LP_128;
            }
            return MI355_GEMM_ALGO_GENERIC;
        }
    }
    // 3 ... 64 rows (or columns): 32 streamed rows x the whole K per workgroup, loader waves, no split-K (gemm_stream64.hip).
    // Interleaved against the split-K 128x128 path over 60 shapes (tools/dev/stream64_probe.py): faster by 5-50 % whenever its
    // grid is at most two rounds of workgroups and not a handful of workgroups each walking a very long K
    // (64 x 8192 x 8192 24.4 us against 29.9, 16 x 8192 x 8192 22.3 / 26.4, 8192 x 64 x 8192 24.4 / 33.7, 32 x 8192 x 2048 7.6 / 14.8;
    // lost: 64 x 32768 x 4096 59 / 45, 64 x 4096 x 16384 38 / 33, 16 x 65536 x 1024 41 / 24).
    // three or four rows (or columns) when the streaming kernel below would not fill the chip (it runs one workgroup per 32
    // streamed rows): the dot-product kernel's many short workgroups beat both MFMA paths (tests/test_gpu_select_audit.py, round 2:
    // 4 x 2048 x 4096 8.8 us against 10.5, 2048 x 4 x 2048 5.7 / 7.1, 384 x 4 x 8192 11.7 / 16.1, 3072 x 4 x 8192 14.6 / 21.4,
    // 2048 x 4 x 14336 21.3 / 27.8, 4096 x 4 x 14336 25.7 / 29.8); from 192 workgroups up the streaming kernel wins
    // (8192 x 4 x 2048 7.0 against 8.2-9.7, 4 x 8192 x 8192 22.0 / 24.8)
    if (std::min(d.m, d.n) > 2 && std::min(d.m, d.n) <= 4 && ((std::max(d.m, d.n) + 31) / 32) * d.batch < 192 && d.batch == 1 &&
        gemm_skinny_supports(d, a, b, c))
        return MI355_GEMM_ALGO_SKINNY;
    if (std::min(d.m, d.n) > 2 && std::min(d.m, d.n) <= 64 && gemm_stream64_supports(d, a, b, c)) {
        const int64_t wgs = ((std::max(d.m, d.n) + 31) / 32) * d.batch, nk64 = d.k / 64;
        const int64_t small_bytes = std::min(d.m, d.n) * d.k * 2;
        // * every workgroup re-reads the small operand from L2: past ~2 MiB the streamed operand evicts it
        //   (64 x 8192 x 28672, 3.5 MiB: 110 us against 93 on the 128x128 path; 8192 x 64 x 14336, 1.75 MiB: 41.4 against 49.0);
        // * up to 32 rows the kernel keeps winning on large grids in its two-workgroups-per-CU form (16 x 28672 x 8192 83.5 us
        //   against 111.5, 16 x 32000 x 4096 42.0 / 53.9); with 33-64 rows only up to two rounds (64 x 14336 x 4096 24.0 / 35.2,
        //   64 x 28672 x 8192 a tie, 64 x 128256 x 4096 295 / 203);
        // * a workgroup walks its K-tiles alone (~0.15 us each): K beyond 8192 needs enough workgroups for that to be hidden
        //   (64 x 4096 x 16384 38 us against 33, 16 x 4096 x 14336 34.5 / 26.7; up to 8192 and 32 rows it wins with any grid: 32 x 512 x 8192
        //   17.8 / 22.2, 512 x 16 x 8192 17.2 / 21.8, 128 x 16 x 8192 17.1 / 18.9).
        // Round 3, after the ring geometry was re-tuned on cold operands (profiles/r03_stream64_ring_geometry.md; 94-shape table
        // profiles/r03_few_rows_audit.txt): any row count walks K up to 8192 alone from 8 workgroups up (48 x 512 x 8192 18.9 us against 23.5, 64 x 512 x 8192
        // 18.9 / 23.1); with 17-32 rows the large grids go to the 128x128 kernel from 768 workgroups (32 x 28672 x 4096 46.7 us
        // against 51.2, 32 x 57344 x 4096 83.2 / 88.2; 32 x 16384 x 8192 at 512 workgroups: 45 against 71 the other way).
        const int64_t rows = std::min(d.m, d.n);
        const int64_t max_wgs = rows <= 16 ? 2048 : rows <= 32 ? 768 : 512;
        // Round 4, after the 128x128 kernel got its 64 x 128 tile (profiles/r04_rows_33_64_ab.txt, cold, us, stream64 / lp128): with
        // 33-64 rows this kernel keeps K <= 2048 on any grid (48 x 4096 x 2048 8.0 / 11.3, 48 x 8192 x 2048 10.5 / 12.6) and the
        // grids of about one workgroup per CU (48 x 8192 x 8192 29.3 / 37.7, 64 x 8192 x 4096 22.1 / 20.5: a tie); fewer
        // workgroups walking a long K lose to split-K (48 x 2048 x 8192 24.4 / 13.3, 64 x 4096 x 8192 29.2 / 19.9, 48 x 4096 x 4096
        // 15.5 / 12.7, 48 x 512 x 8192 19.1 / 10.2), and so do more workgroups than CUs (64 x 10240 x 4096 34.2 / 24.2, 64 x 14336 x 4096
        // 36.2 / 33.2), K = 4096 on a full grid (64 x 8192 x 4096 22.1 / 20.5) and K past 8192 (64 x 8192 x 14336 57.7 / 44.0).
        // Up to 32 rows the same holds with wider margins (profiles/r04_rows_le32_ab.txt): K = 8192 on fewer than 192 workgroups goes
        // to split-K (32 x 512 x 8192 16.3 / 9.5, 16 x 2048 x 8192 18.8 / 11.9, 32 x 4096 x 8192 22.4 / 18.4; 32 x 6144 x 8192 24.5 / 26.4 the
        // other way), K = 4096 is a tie from 96 workgroups up and 7 % behind below.
        // Few COLUMNS are not the mirror image: the 128x128 kernel has its 64 x 128 tile for few ROWS only, so with the large
        // operand on the A side the streaming kernel keeps K <= 4096 on short grids (1472 x 25 x 3072 8.4 / 13.3, 8192 x 48 x 4096
        // 20.9 / 22.4) -- and loses K past 8192 even on a full grid (9312 x 24 x 14336 68.9 / 53.8, 8192 x 64 x 14336 59.3 / 52.1;
        // 32 x 8192 x 16384 with the rows on the A side: 48.1 / 62.0 the other way).
        const bool few_rows = d.m <= d.n;
        const bool grid_ok = rows <= 32 ? (nk64 <= 32 || (nk64 <= 64 && (wgs >= 96 || !few_rows)) || (wgs >= 192 && (few_rows || nk64 <= 128)))
                                        : (nk64 <= 32 || (wgs >= 224 && wgs <= 288 && nk64 > 64 && nk64 <= 128));
        // ... and with K <= 4096 few columns stream on any grid (the workgroup limits above were measured with the rows on the A
        // side): 37824 x 23 x 3072 44.8 / 69.8, 51016 x 17 x 1536 32.9 / 39.1 (tools/dev/random_audit.py).
        // (up to 32 columns: with 64 the tile kernels win there -- 51880 x 64 x 1024 42.2 / 26.2, 20880 x 64 x 1024 19.0 / 14.9.)
        // Few rows whose 128-column tiles would fill a second round by less than three quarters also stay here past the
        // workgroup limit: 20 x 40096 x 8192 (314 tiles) 100.4 / 151.1.
        const int64_t t128 = (std::max(d.m, d.n) + 127) / 128 * d.batch;
        const bool any_grid = (!few_rows && rows <= 32 && nk64 <= 64) || (few_rows && rows <= 32 && t128 > 256 && t128 < 448 && nk64 >= 64);
        if (small_bytes <= (1ll << 21) && (any_grid || (wgs <= max_wgs && grid_ok))) return MI355_GEMM_ALGO_STREAM64;
    }
    // one or two rows (or columns): HBM-bound on the other operand; stream it once with dot products, no MFMA tile to fill
    // (gemm_skinny.hip: 20.4 us against 24.7 at 1 x 8192 x 8192).  Up to 16 rows when no MFMA kernel takes the descriptor.
    if ((d.m <= 16 || d.n <= 16) && gemm_skinny_supports(d, a, b, c) && (std::min(d.m, d.n) <= 2 || !(big4 || mid)))
        return MI355_GEMM_ALGO_SKINNY;
    // Output-bound (K of at most four K-tiles) over more than one round of 256x256 tiles: the 128x128 kernel in its
    // single-stage form, four workgroups per CU -- the C stores of three hide the fetch + MFMA phase of the fourth, where a
    // 256x256 workgroup (alone on its CU) serialises them.  bf16, 8192 x 8192 x K (tools/dev/shortk_probe.py): K = 64 32.3 us
    // against 41.9, K = 128 37.1 against 38.9 (persistent), K = 192 40.9 / 43.0, K = 256 46.8 / 49.0; 16384 x 8192 x 64
    // 51.6 / 68.8.  A single round (4096 x 4096: 256 tiles) stays with the large tile (10.6 us against 12.2).
    // Round 3, cold operands (launches rotating through operand sets larger than the Infinity Cache, tools/ab_algos.py).  With the
    // single-stage form's C stores non-temporal (gemm_lp128.hip LP128_NT; with plain stores the window had shrunk to K <= 128 on
    // cold buffers) it wins up to K = 192 everywhere measured and up to K = 256 on grids of at most 1280 tiles of 256x256
    // (8192 x 8192 x 192: 42.1 us against 48.2 persistent, x 256: 47.5 / 54.7, 8192 x 4096 x 256: 26.1 / 28.7, 12288 x 8192 x 192: 60.6 / 67.0;
    // 16384 x 8192 x 256: 96.1 / 90.5 the other way, 8192 x 6144 x 256 a tie; K = 320 is past the single-stage form: 68 / 59;
    // profiles/r03_short_k_cold.txt, r03_short_k_cold_nt_stores.txt).
    {
        const int64_t tiles256 = ((d.m + 255) / 256) * ((d.n + 255) / 256) * d.batch;
        if (mid && tiles256 > 256 && (d.k <= 192 || (d.k <= 256 && tiles256 <= 1280))) return MI355_GEMM_ALGO_LP_128;
    }
    // at most 128 rows (or columns) over many tiles: a 256-row tile multiplies at least half zeros and streams no faster --
    // 64 x 128256 x 4096: 192 us on the 128x128 kernel against 222
    // (few COLUMNS fill the 256 rows of the tall tile: one round of it, where it exists, is the exception -- see its rule below)
    if (mid && std::min(d.m, d.n) <= 128) {
        const int64_t t128 = ((d.m + 127) / 128) * d.batch, tall = ((d.m + 255) / 256) * d.batch;
        if (d.n <= 128 && d.m > 128 && t128 > 256 && tall <= 256 && gemm_lp256x128_supports(d, a, b, c)) return MI355_GEMM_ALGO_LP_256X128;
        return MI355_GEMM_ALGO_LP_128;
    }
    // Round 5: ONE round of 192 x 192 tiles of the 4-wave kernel (gemm_lp256w4.hip NJ = NI = 3) that fills most of the chip -- the
    // mid-size band where the square 256 tile leaves >= 44 % of the CUs idle and the 128-wide tiles are bound by L2 -> LDS delivery.
    // A K-tile of it costs ~0.85 of the square tile's for 0.56 of the work, so it pays exactly while it stays one round
    // (profiles/r05_tile_192x192_ab.txt, cold, TFLOP/s, against the best of 256x256 / 256x192 / 256x128 / 128x128): 3072^3 1185 / 1105,
    // 3072^2 x 8192 1257 / 1172, 2304^3 859 / 716, 2688^3 989 / 913, 2816^2 x 4096 1085 / 1021, 3072 x 2304 x 4096 994 / 943,
    // 4096 x 2048 x 4096 1131 / 1068, 2560^2 x 4096 915 / 894; below ~144 tiles the 128x128 kernel's two workgroups per CU win
    // (2048^3: 121 tiles, 696 / 762; 1920^2 x 4096: 100 tiles, 720 / 777), a long K on few tiles is a tie with the 256 x 128 tile
    // (3072 x 1536 x 8192: 128 tiles, 727 / 732; 2048 x 3072 x 8192: 176 tiles, 913 / 941 -- kept on that tile).
    // (K of at least 16 K-tiles: 3072^2 x 512 631 on the square tile against 525)
    if (big4 && d.trans_b && std::min(d.m, d.n) > 512 && d.k >= 1024 && gemm_lp256x192_supports(d, a, b, c)) {
        const int64_t tiles192 = ((d.m + 191) / 192) * ((d.n + 191) / 192) * d.batch;
        if (tiles192 >= 144 && tiles192 <= 256 && !(d.k >= 8192 && tiles192 < 192)) return MI355_GEMM_ALGO_LP_192X192;
    }
    // More than one 128x128 tile per CU but at most one 256x128 tile per CU, and a long K: the 256 x 128 form of the same
    // kernel (gemm_lp128.hip, MI = 4) -- 0.75 x the L2 -> LDS bytes per FLOP, which is what the two co-resident 128x128
    // workgroups per CU are bound by.  Interleaved, cold operands (profiles/r03_tile_256x128_sweep.txt, 50 shapes): K >= 3072
    // +7 ... +24 % (4096 x 2048 x 4096 72.1 -> 62.2 us, 2560^2 x 4096 67.1 -> 56.4, 4096 x 1536 x 4096 68.5 -> 55.1), K = 2048 a
    // tie, K = 1024 -3 ... -7 %; with at most one 128x128 tile per CU it loses (2048^3: 27.5 us against 19.0 -- half the CUs).
    {
        const int64_t t128 = ((d.m + 127) / 128) * ((d.n + 127) / 128) * d.batch, tall = ((d.m + 255) / 256) * ((d.n + 127) / 128) * d.batch;
        // Round 4 (profiles/r04_tall_skinny_ab.txt): with one side of at most 512 the short-K exclusion does not hold -- the 256-row
        // tile halves the workgroups that re-read the small operand: 44440 x 88 x 1536 43.3 -> 35.7 us, 51880 x 64 x 1024 28.3 -> 24.1,
        // 16384 x 512 x 1024 26.2 -> 21.4, 32768 x 256 x 1024 28.0 -> 23.0, 40000 x 96 x 512 16.6 -> 14.9 (65536 x 64 x 512: a tie).
        if (mid && t128 > 256 && tall <= 256 && (d.k >= 2560 || std::min(d.m, d.n) <= 512) && gemm_lp256x128_supports(d, a, b, c))
            return MI355_GEMM_ALGO_LP_256X128;
    }
    if (big4) {
        // 256x256 tiles once the 128x128 kernel would need more than its two co-resident workgroups per CU (512 tiles of
        // 128^2 = 128 of 256^2).  Measured (tools/ab_algos.py): 96-128 tiles a tie, 144-160 tiles +45...55 % for the
        // 256x256 kernel even though it leaves 40 % of the CUs idle, 64-81 tiles +7...30 % for the 128x128 kernel.
        const int64_t tiles256 = ((d.m + 255) / 256) * ((d.n + 255) / 256) * d.batch;
        // Round 5: one round of 256x256 tiles that leaves CUs idle, and a round of 256 x 192 tiles (the same kernel, NJ = 3) that
        // fills more of them.  A K-tile of the narrower tile costs 0.86-0.90 of the square one's (12 MFMAs per k-step against 16 on
        // the same A fragments), so it pays exactly when it stays ONE round (profiles/r05_tile_256x192_ab.txt, cold, TFLOP/s,
        // 256x256 / 256x192): 3072^3 917 / 1072, 3072 x 3072 x 8192 1052 / 1168, 4096 x 3072 x 4096 1194 / 1299, 3328^2 x 4096 1157 / 1246;
        // two rounds lose: 3584^3 1146 / 788, 3072 x 4096 x 4096 1194 / 812, 4096^3 1313 / 979.
        if (big4 && tiles256 > 128 && tiles256 < 256 && d.trans_b && gemm_lp256x192_supports(d, a, b, c)) {
            const int64_t tiles192 = ((d.m + 255) / 256) * ((d.n + 191) / 192) * d.batch;
            if (tiles192 <= 256 && tiles192 > tiles256) return MI355_GEMM_ALGO_LP_256X192;
        }
        if (tiles256 > 128 || !mid) return MI355_GEMM_ALGO_LP_256W4;
    }
    if (mid) return MI355_GEMM_ALGO_LP_128;
    return MI355_GEMM_ALGO_GENERIC;
}

}  // namespace

namespace {

// When AUTO would land on the generic scalar kernel because of an operand's LAYOUT (transposed A, 16-bit row-major
// B), its K extent (not a multiple of the K-tile) or its alignment (rows / base not 16-byte aligned), the operand is
// first re-laid out into K-contiguous, zero-padded, aligned library scratch -- what the reference's launchers do
// with into_contiguous after matrix_batch_layout -- and the MFMA kernel runs on that.  `plan_relayout` is the pure
// part (also used by mi355_gemm_select); it returns false when nothing would be gained.
struct relayout_plan {
    bool a, b;              // which operands get a scratch copy
    int64_t kpad;           // K rounded up to the K-tile (zero columns add nothing)
    mi355_gemm_desc nd;     // the descriptor the MFMA kernel sees
};

bool plan_relayout(const mi355_gemm_desc &d, const void *a, const void *b, const void *c, relayout_plan &p)
{
    if (d.k == 0 || d.m * d.n * d.k < (int64_t)1 << 21) return false;           // tiny: not worth extra launches
    const int64_t esz = (int64_t)dtype_size(d.dtype_ab);
    const int64_t ktile = 128 / esz, ve = 16 / esz;
    p.kpad = (d.k + ktile - 1) / ktile * ktile;
    const bool ragged_k = p.kpad != d.k;
    const bool a_misaligned = (d.lda % ve) || (d.stride_a % ve) || (reinterpret_cast<uintptr_t>(a) & 15u);
    const bool b_misaligned = (d.ldb % ve) || (d.stride_b % ve) || (reinterpret_cast<uintptr_t>(b) & 15u);
    p.a = d.trans_a != 0 || ragged_k || a_misaligned;
    p.b = (!d.trans_b && (d.dtype_ab != MI355_DTYPE_F32 || ragged_k || b_misaligned || (d.n & 3))) ||
          (d.trans_b && (ragged_k || b_misaligned));
    if (!p.a && !p.b) return false;
    static const char aligned_dummy __attribute__((aligned(16))) = 0;
    auto planned = [&](bool pa, bool pb) {
        p.a = pa; p.b = pb;
        p.nd = d;
        p.nd.k = p.kpad;
        if (p.a) { p.nd.trans_a = 0; p.nd.lda = p.kpad; p.nd.stride_a = d.stride_a == 0 ? 0 : d.m * p.kpad; }
        if (p.b) { p.nd.trans_b = 1; p.nd.ldb = p.kpad; p.nd.stride_b = d.stride_b == 0 ? 0 : d.n * p.kpad; }
        return select(p.nd, p.a ? &aligned_dummy : a, p.b ? &aligned_dummy : b, c) != MI355_GEMM_ALGO_GENERIC;
    };
    // A 16-bit row-major B that is only here because A needs a new layout stays where it is when a tile kernel stages it natively
    // (until late round 3 both operands were transposed: 2048 x 2048 x 8192 with A stored [K][M] 110.6 us against 95.5 with B left alone)
    if (p.a && p.b && !d.trans_b && !ragged_k && !b_misaligned && d.dtype_ab != MI355_DTYPE_F32 && planned(true, false)) return true;
    return planned(p.a, p.b);
}

int32_t relayout_for_mfma(mi355_ctx *ctx, hipStream_t s, const mi355_gemm_desc &d, const void *a, const void *b, void *c,
                          mi355_gemm_desc &nd, const void *&na, const void *&nb)
{
    relayout_plan p;
    if (!plan_relayout(d, a, b, c, p)) return MI355_E_UNSUPPORTED;
    const int esz = (int)dtype_size(d.dtype_ab);
    nd = p.nd; na = a; nb = b;
    if (p.a) {
        const int64_t nba = d.stride_a == 0 ? 1 : d.batch;
        void *q = nullptr;
        if (scratch_get(ctx, s, SCRATCH_RELAYOUT_A, (size_t)(nba * d.m * p.kpad * esz), &q) != MI355_OK) return MI355_E_UNSUPPORTED;
        if (d.trans_a) {                                                       // [K][M] -> [M][kpad], pad columns zeroed first
            if (p.kpad != d.k) (void)hipMemsetAsync(q, 0, (size_t)(nba * d.m * p.kpad * esz), s);
            launch_transpose(s, a, q, d.k, d.m, d.lda, p.kpad, nba, d.stride_a, d.m * p.kpad, esz);
        } else {
            launch_pad_copy(s, a, q, d.m, d.k, p.kpad, d.lda, p.kpad, nba, d.stride_a, d.m * p.kpad, esz);
        }
        na = q;
    }
    if (p.b) {
        const int64_t nbb = d.stride_b == 0 ? 1 : d.batch;
        void *q = nullptr;
        if (scratch_get(ctx, s, SCRATCH_RELAYOUT_B, (size_t)(nbb * d.n * p.kpad * esz), &q) != MI355_OK) return MI355_E_UNSUPPORTED;
        if (!d.trans_b) {                                                      // [K][N] -> [N][kpad]
            if (p.kpad != d.k) (void)hipMemsetAsync(q, 0, (size_t)(nbb * d.n * p.kpad * esz), s);
            launch_transpose(s, b, q, d.k, d.n, d.ldb, p.kpad, nbb, d.stride_b, d.n * p.kpad, esz);
        } else {
            launch_pad_copy(s, b, q, d.n, d.k, p.kpad, d.ldb, p.kpad, nbb, d.stride_b, d.n * p.kpad, esz);
        }
        nb = q;
    }
    check_launch(ctx, "mi355_gemm(operand re-layout)");
    return MI355_OK;
}

// ---- the last, partly filled round of 256x256 tiles ----------------------------------------------------------------------
// All tiles of a GEMM take the same time, so T tiles on 256 CUs cost ceil(T / 256) rounds: 576 tiles (6144^3) pay for
// 768.  When the leftover is at most half a round, the output is cut into a part whose tile count (nearly) fills whole
// rounds and a strip of at most 96 tiles whose K range is split `s` ways over the CUs that would otherwise idle: the
// strip runs as ONE batched launch of the same kernel (batch entry z = K slice z: operand "batch strides" of one slice
// along the rows, f32 partial slabs as the batched output) followed by the split-K fold (gemm_splitk.hip), which adds
// the slabs in slice order (deterministic) and converts.  Chosen by a cost model in K-tile units; only for batch == 1.
struct tail_plan {
    bool along_m;          // the strip is the last rows (true) or the last columns (false) of C
    int64_t main_extent;   // rows (columns) handled by the plain launch
    int64_t splits;
};

bool plan_tail_split(const mi355_gemm_desc &d, tail_plan &best)
{
    if (d.batch != 1 || d.trans_a || (d.n & 3)) return false;
    const int64_t esz = (int64_t)dtype_size(d.dtype_ab), ktile = 128 / esz, nk = d.k / ktile;
    if (!d.trans_b && ((d.n & (16 / esz - 1)) || is_fp8(d.dtype_ab))) return false;   // row-major B: f32 and 16-bit, whole 16-byte pieces
    const int64_t tm = (d.m + 255) / 256, tn = (d.n + 255) / 256, T = tm * tn;
    if (nk < 8) return false;
    // cycles: a K-tile costs ~2 200, a tile's prologue + epilogue ~10 000, a launch ~5 000, the fold moves
    // (s + 1) x 256 KiB per strip tile at ~2 500 B per cycle of the whole chip
    const double CK = 2200.0, CFIX = 10000.0, CLAUNCH = 5000.0;
    // A partly filled round is cheaper than a full one: the idle CUs' power budget lets the busy ones clock higher
    // (64 leftover tiles cost ~0.66 of a round, 128 ~0.83: 6144^3 and 4096 x 6144 x 4096).  Calibrated interleaved against
    // the plain launch (tools/dev/tail_ab.py): with F0 = 0.7 and strips of at most 96 tiles the split is taken at
    // 1.02-1.13, 2.25, 3.06 and 4.12 rounds (+28, +18, +15, +11, +7, +7, +7 %) and left alone at 1.5-1.56, 2.44, 4.5 rounds.
    constexpr double F0 = 0.7;
    auto rounds = [](int64_t t) {
        const int64_t left = t % 256;
        return (double)(t / 256) + (left ? F0 + (1.0 - F0) * (double)left / 256.0 : 0.0);
    };
    const double now = rounds(T) * ((double)nk * CK + CFIX);
    double best_cost = now * 0.93;                          // must win by 7 % to be worth two extra launches
    bool found = false;
    for (int dir = 0; dir < 2; ++dir) {
        const int64_t t_along = dir == 0 ? tm : tn, t_other = dir == 0 ? tn : tm;
        // the strip is whole tiles of the LAST rows (columns): when that extent ends in a partly filled tile the model's tile
        // count is wrong and the split lost (4160 x 10240 x 8192: 618 us against 531 plain, tools/dev/random_audit.py)
        if ((dir == 0 ? d.m : d.n) % 256) continue;
        for (int64_t strip = 1; strip <= t_along && strip * t_other <= 96; ++strip) {
            const int64_t t_strip = strip * t_other, t_main = T - t_strip;
            for (int64_t sp = 2; sp <= 16 && sp * t_strip <= 256 && nk / sp >= 4; ++sp) {
                if (nk % sp) continue;
                const double fold = (double)t_strip * (double)(sp + 1) * 262144.0 / 2500.0;
                const double cost = rounds(t_main) * ((double)nk * CK + CFIX) + rounds(sp * t_strip) * ((double)(nk / sp) * CK + CFIX) + fold +
                                    (t_main > 0 ? 2.0 : 1.0) * CLAUNCH;
                if (cost < best_cost) {
                    best_cost = cost;
                    best.along_m = dir == 0;
                    best.main_extent = (t_along - strip) * 256;
                    best.splits = sp;
                    found = true;
                }
            }
        }
    }
    return found;
}

// Several rounds of short tiles: the persistent form (gemm_lp256p.hip) streams the next tile's operands during the epilogue
// and lets the C stores drain under the next tile's MFMAs.  Measured against the one-tile-per-workgroup kernel,
// interleaved (tools/dev/p_vs_w4.py): K = 512 +10 %, 1024 +4 %, 2048 +1...4 %, 4096 +1 %, 8192 a tie; single-round
// launches -0.5...-2 %.  A leftover round that the strip split handles gains more from that and keeps the plain kernel.
bool prefers_persistent(const mi355_gemm_desc &d, const void *a, const void *b, const void *c)
{
    if (d.dtype_ab != MI355_DTYPE_BF16 && d.dtype_ab != MI355_DTYPE_F16) return false;
    const int64_t tiles = ((d.m + 255) / 256) * ((d.n + 255) / 256) * d.batch, nk = d.k / 64;
    // more than one round of tiles (round 2: was two full rounds).  With 257-511 tiles the second round is partly empty either
    // way, and the tile hand-over of the persistent forms still hides the epilogues: 8192 x 3072 x 512 (384 tiles) 32.5 ->
    // 28.1 us, x 1024 50.0 -> 45.8, 7168 x 4096 x 1024 (448) 52.1 -> 49.9, 8192 x 2560 x 512 (320) 30.2 -> 27.2, level from
    // K = 2048 up (tools/dev/shortk_probe.py)
    if (tiles <= 256 || nk > 64) return false;
    if (!gemm_lp256p_supports(d, a, b, c)) return false;
    tail_plan tp;
    return !plan_tail_split(d, tp);
}

// 16-bit C: the persistent kernel that keeps the finished tile in registers and drips its stores into the next tile's K loop
// (gemm_lp256q.hip).  Measured against gemm_lp256p.hip, interleaved, 64 x 2048 x 2048 x K (tools/dev/q_ab.py): K = 384 / 448 /
// 512 (8 stores per K-tile) +1.5 / +2.5 / +4 %, K = 576 ... 960 (4 or 2 stores per K-tile) -5 ... -11 %, K = 1024 / 1536 / 2048 /
// 4096 +3.6 / +1.5 / +3.6 ... 4.9 / +1.2 %, 8192 a tie.
bool prefers_dripped_stores(const mi355_gemm_desc &d, const void *a, const void *b, const void *c)
{
    const int64_t nk = d.k / 64;
    return gemm_lp256q_supports(d, a, b, c) && (nk >= 16 || (nk >= 6 && nk <= 8));
}

int32_t select_auto(const mi355_gemm_desc &d, const void *a, const void *b, const void *c, bool strip_kernel = true)
{
    const int32_t algo = select(d, a, b, c, strip_kernel);
    if (algo != MI355_GEMM_ALGO_LP_256W4) return algo;
    if (prefers_persistent(d, a, b, c)) return prefers_dripped_stores(d, a, b, c) ? MI355_GEMM_ALGO_LP_256Q : MI355_GEMM_ALGO_LP_256P;
    // Round 5: several full rounds of 256x256 tiles of [N][K] 16-bit operands run at the chip's POWER limit on random data; there the
    // same tile on v_mfma_f32_16x16x32 (gemm_lp256m16.hip: half the accumulator bytes through the register file per FLOP) holds
    // 1.83-1.87 GHz where the 32x32x16 kernel holds 1.63: 8192^3 1 440 -> 1 533-1 563 TFLOP/s.  Below ~3 rounds the chip is not
    // power-bound and the wider instruction's slack per issue slot wins (4096^3 a tie, 3584^3 -4 %): profiles/r05_m16_ab.txt.
    // (Not where the launcher would split a ragged last round off: that planner's model is the 32x32 kernel's.)
    {
        const int64_t tiles256 = ((d.m + 255) / 256) * ((d.n + 255) / 256) * d.batch;
        tail_plan tp;
        if (tiles256 >= 768 && gemm_lp256m16_supports(d, a, b, c) && !plan_tail_split(d, tp)) return MI355_GEMM_ALGO_LP_256M16;
    }
    return algo;
}

int32_t run_tail_split(mi355_ctx *ctx, hipStream_t s, const mi355_gemm_desc &d, const void *a, const void *b, void *c,
                       const tail_plan &p)
{
    const int64_t esz = (int64_t)dtype_size(d.dtype_ab), csz = (int64_t)dtype_size(d.dtype_c);
    const int64_t ms = p.along_m ? d.m - p.main_extent : d.m, ns = p.along_m ? d.n : d.n - p.main_extent;   // the strip
    float *slabs = nullptr;
    if (splitk_scratch(ctx, s, (size_t)(p.splits * ms * ns) * sizeof(float), &slabs) != MI355_OK) return MI355_E_UNSUPPORTED;
    const char *a0 = static_cast<const char *>(a), *b0 = static_cast<const char *>(b);
    char *c0 = static_cast<char *>(c);
    const char *as = p.along_m ? a0 + p.main_extent * d.lda * esz : a0;
    // the strip's first column of B: a row of [N][K] storage, a column of row-major [K][N]
    const char *bs = p.along_m ? b0 : b0 + (d.trans_b ? p.main_extent * d.ldb * esz : p.main_extent * esz);
    char *cs = p.along_m ? c0 + p.main_extent * d.ldc * csz : c0 + p.main_extent * csz;
    mi355_gemm_desc sd = d;                                 // the strip: batch entry z = K slice z, f32 slab output
    sd.m = ms; sd.n = ns; sd.k = d.k / p.splits; sd.batch = p.splits;
    sd.stride_a = sd.k; sd.stride_b = d.trans_b ? sd.k : sd.k * d.ldb; sd.stride_c = ms * ns; sd.ldc = ns; sd.dtype_c = MI355_DTYPE_F32;
    if (!gemm_lp256w4_supports(sd, as, bs, slabs)) return MI355_E_UNSUPPORTED;
    if (p.main_extent > 0) {
        mi355_gemm_desc md = d;
        if (p.along_m) md.m = p.main_extent; else md.n = p.main_extent;
        if (!gemm_lp256w4_supports(md, a, b, c)) return MI355_E_UNSUPPORTED;
        const int32_t rc = launch_gemm_lp256w4(ctx, s, md, a, b, c);
        if (rc != MI355_OK) return rc;
    }
    const int32_t rc = launch_gemm_lp256w4(ctx, s, sd, as, bs, slabs);
    if (rc != MI355_OK) return rc;
    launch_splitk_fold(s, slabs, (uint32_t)p.splits, ms * ns, 1, ms, ns, cs, d.dtype_c, d.ldc, 0);
    check_launch(ctx, "mi355_gemm(tail split-K fold)");
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
    int32_t algo = d.algo == MI355_GEMM_ALGO_AUTO ? select_auto(d, a, b, c) : d.algo;
    if (d.algo == MI355_GEMM_ALGO_AUTO && algo == MI355_GEMM_ALGO_GENERIC) {
        mi355_gemm_desc nd;
        const void *na, *nb;
        if (relayout_for_mfma(ctx, s, d, a, b, c, nd, na, nb) == MI355_OK) {
            d = nd; a = na; b = nb;
            algo = select_auto(d, a, b, c);
        }
    }
    // the strip kernel needs scratch + ticket words that cannot be created inside a capture window: AUTO then takes what it took
    // before that kernel existed (as the split-K paths fall back to their unsplit forms)
    // -- through the same AUTO pipeline (persistent-kernel promotion; GENERIC = "re-lay out, then select again"), so that a
    // captured graph replays the kernel AUTO would have taken, not the scalar correctness net
    if (d.algo == MI355_GEMM_ALGO_AUTO && algo == MI355_GEMM_ALGO_NNROWS && !gemm_nnrows_ready(ctx, s, d)) {
        algo = select_auto(d, a, b, c, false);
        if (algo == MI355_GEMM_ALGO_GENERIC) {
            mi355_gemm_desc nd;
            const void *na, *nb;
            if (relayout_for_mfma(ctx, s, d, a, b, c, nd, na, nb) == MI355_OK) {
                d = nd; a = na; b = nb;
                algo = select_auto(d, a, b, c, false);
            }
        }
    }
    if (d.algo == MI355_GEMM_ALGO_AUTO && algo == MI355_GEMM_ALGO_LP_256W4) {
        tail_plan tp;
        if (plan_tail_split(d, tp) && run_tail_split(ctx, s, d, a, b, c, tp) == MI355_OK) return MI355_OK;
    }
    switch (algo) {
    case MI355_GEMM_ALGO_GENERIC: return launch_gemm_generic(ctx, s, d, a, b, c);
    case MI355_GEMM_ALGO_F32_MFMA: return launch_gemm_f32_mfma(ctx, s, d, a, b, c);
    case MI355_GEMM_ALGO_LP_128: return launch_gemm_lp128(ctx, s, d, a, b, c);
    case MI355_GEMM_ALGO_LP_256: return launch_gemm_lp256w4(ctx, s, d, a, b, c);      // retired 8-wave kernel: an alias since ABI 8
    case MI355_GEMM_ALGO_LP_256W4: return launch_gemm_lp256w4(ctx, s, d, a, b, c);
    case MI355_GEMM_ALGO_LP_256P: return launch_gemm_lp256p(ctx, s, d, a, b, c);
    case MI355_GEMM_ALGO_LP_256Q: return launch_gemm_lp256q(ctx, s, d, a, b, c);
    case MI355_GEMM_ALGO_SKINNY: return launch_gemm_skinny(ctx, s, d, a, b, c);
    case MI355_GEMM_ALGO_STREAM64: return launch_gemm_stream64(ctx, s, d, a, b, c);
    case MI355_GEMM_ALGO_LP_256X128: return launch_gemm_lp256x128(ctx, s, d, a, b, c);
    case MI355_GEMM_ALGO_NNROWS: return launch_gemm_nnrows(ctx, s, d, a, b, c);
    case MI355_GEMM_ALGO_LP_256X192: return launch_gemm_lp256x192(ctx, s, d, a, b, c);
    case MI355_GEMM_ALGO_LP_192X192: return launch_gemm_lp256x192(ctx, s, d, a, b, c, 192);
    case MI355_GEMM_ALGO_LP_256M16: return launch_gemm_lp256m16(ctx, s, d, a, b, c);
    default: return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_gemm: unknown algo %d", algo);
    }
}

// cmma::execute(a, b, c, d) at tensor level: D = A * B + C.  The selected kernel writes the f32 product into
// library-owned per-stream scratch; gemm_add.hip adds C in f32 and rounds once to dtype_c.
MI355_API int32_t mi355_gemm_add(mi355_ctx *ctx, mi355_stream stream, const mi355_gemm_desc *desc, const void *a, const void *b,
                                 const void *c, void *d_out)
{
    MI355_REQUIRE_CTX(ctx);
    int32_t rc = validate(ctx, desc, a, b, d_out);
    if (rc != -1) return rc;
    if (!c) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_gemm_add: the C operand is NULL");
    hipStream_t s = stream_of(ctx, stream);
    const mi355_gemm_desc &d = *desc;
    // f32 output on the 256x256 kernel: C is added inside the kernel's epilogue (one extra read of C, nothing else)
    if (d.dtype_c == MI355_DTYPE_F32 && (d.algo == MI355_GEMM_ALGO_AUTO || d.algo == MI355_GEMM_ALGO_LP_256W4) &&
        (reinterpret_cast<uintptr_t>(c) & 15u) == 0 && gemm_lp256w4_supports(d, a, b, d_out) &&
        (d.algo == MI355_GEMM_ALGO_LP_256W4 || select_auto(d, a, b, d_out) == MI355_GEMM_ALGO_LP_256W4 ||
         select_auto(d, a, b, d_out) == MI355_GEMM_ALGO_LP_256P))
        return launch_gemm_lp256w4(ctx, s, d, a, b, d_out, c);
    void *prod = nullptr;
    const size_t bytes = (size_t)d.batch * (size_t)d.m * (size_t)d.n * sizeof(float);
    if (scratch_get(ctx, s, SCRATCH_PRODUCT, bytes, &prod) != MI355_OK)
        return fail(ctx, MI355_E_UNSUPPORTED, "mi355_gemm_add: no scratch for the %zu-byte f32 product (inside a capture window?)", bytes);
    mi355_gemm_desc pd = d;
    pd.dtype_c = MI355_DTYPE_F32;
    pd.ldc = d.n;
    pd.stride_c = d.m * d.n;
    rc = mi355_gemm(ctx, stream, &pd, a, b, prod);
    if (rc != MI355_OK) return rc;
    launch_add_c(s, static_cast<const float *>(prod), c, d_out, d.batch, d.m, d.n, d.dtype_c, d.ldc, d.stride_c);
    check_launch(ctx, "mi355_gemm_add");
    return MI355_OK;
}

// Introspection (no device needed): how mi355_gemm would cut a descriptor that AUTO resolves to the 256x256 kernel.
MI355_API int32_t mi355_gemm_split_plan(const mi355_gemm_desc *desc, int32_t compute_units, int32_t *out_slices)
{
    if (!desc || !out_slices || compute_units < 0) return MI355_E_INVALID_ARGUMENT;
    static const char aligned_dummy __attribute__((aligned(16))) = 0;
    *out_slices = 1;
    if (desc->m <= 0 || desc->n <= 0 || desc->k <= 0 || desc->batch <= 0) return MI355_OK;
    if (!gemm_lp128_supports(*desc, &aligned_dummy, &aligned_dummy, &aligned_dummy)) return MI355_OK;   // the launcher would never see it
    *out_slices = (int32_t)lp128_split_count(*desc, compute_units ? compute_units : 256);
    return MI355_OK;
}

MI355_API int32_t mi355_gemm_tail_plan(const mi355_gemm_desc *desc, int32_t *out_along_m, int64_t *out_main_extent, int32_t *out_splits)
{
    if (!desc || !out_along_m || !out_main_extent || !out_splits) return MI355_E_INVALID_ARGUMENT;
    tail_plan tp{};
    const bool split = plan_tail_split(*desc, tp);
    *out_along_m = split ? (tp.along_m ? 1 : 0) : 0;
    *out_main_extent = split ? tp.main_extent : 0;
    *out_splits = split ? (int32_t)tp.splits : 1;
    return MI355_OK;
}

MI355_API int32_t mi355_gemm_relayout_plan(const mi355_gemm_desc *desc, int32_t *out_relayout_a, int32_t *out_relayout_b)
{
    if (!desc || !out_relayout_a || !out_relayout_b) return MI355_E_INVALID_ARGUMENT;
    static const char aligned_dummy __attribute__((aligned(16))) = 0;
    *out_relayout_a = *out_relayout_b = 0;
    if (desc->m <= 0 || desc->n <= 0 || desc->batch <= 0) return MI355_OK;
    if (select_auto(*desc, &aligned_dummy, &aligned_dummy, &aligned_dummy) != MI355_GEMM_ALGO_GENERIC) return MI355_OK;
    relayout_plan p;                  // exactly what mi355_gemm does before it settles for the scalar kernel
    if (plan_relayout(*desc, &aligned_dummy, &aligned_dummy, &aligned_dummy, p)) {
        *out_relayout_a = p.a ? 1 : 0;
        *out_relayout_b = p.b ? 1 : 0;
    }
    return MI355_OK;
}

MI355_API int32_t mi355_gemm_select(mi355_ctx *ctx, const mi355_gemm_desc *desc, int32_t *out_algo)
{
    (void)ctx;                           // may be NULL: nothing below asks the device
    if (!desc || !out_algo) return MI355_E_INVALID_ARGUMENT;
    // alignment-dependent choices are evaluated for 16-byte aligned operands
    static const char aligned_dummy __attribute__((aligned(16))) = 0;
    const mi355_gemm_desc &d = *desc;
    int32_t algo = select_auto(d, &aligned_dummy, &aligned_dummy, &aligned_dummy);
    if (algo == MI355_GEMM_ALGO_GENERIC) {
        relayout_plan p;                  // what mi355_gemm does before it settles for the scalar kernel
        if (plan_relayout(d, &aligned_dummy, &aligned_dummy, &aligned_dummy, p))
            algo = select_auto(p.nd, &aligned_dummy, &aligned_dummy, &aligned_dummy);
    }
    *out_algo = algo;
    return MI355_OK;
}
