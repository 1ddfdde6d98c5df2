// gemm_f32.hip -- f32 GEMM on the gfx950 f32-input matrix core (v_mfma_f32_32x32x2_f32).
//
// Roofline: MFMA f32, 157.3 TFLOP/s (64 FLOP/clk/SIMD; exact f32, each product rounded once, an
// fmaf chain -- cdna_hip_programming.md section 3 "FP32-input MFMA").  4096^3: 1.374e11 FLOP,
// 201 MB minimum traffic => compute-bound by 30x.
//
// Structure: 128x128x32 workgroup tile, 4 waves (2x2), each wave a 64x64 output = 2x2 MFMA tiles
// (64 accumulator registers).  Global -> registers -> LDS double buffer, one barrier per K-tile,
// the next tile's global loads issued before the current tile's 64 MFMAs (4096 matrix-pipe
// cycles per wave: covers HBM latency several times over).
//
// K permutation: a 32x32x2 MFMA consumes ONE k value per lane-half.  Instead of reading one f32
// per MFMA, each lane reads 4 consecutive k (ds_read_b128) and feeds element j to the j-th of four
// MFMAs: lane-half h then supplies k = 8t + 4h + j.  A and B use the same map, so every k in
// [8t, 8t+8) is used exactly once -- only the order of the f32 accumulation changes.
//
// Operands are SWAPPED in the MFMA call (first = B fragment, second = A fragment) so that each
// lane ends up with 4 consecutive N-columns of one C row per register quad => 16-byte stores.
#include "gemm_common.hpp"

using namespace mi355;

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDK = BK + 4;   // 144-byte row pitch: conflict-free ds_read_b128 over 16-lane groups
constexpr int LDN = BN + 4;   // row-major-B image [BK][BN+4]

struct __attribute__((aligned(16))) f32_smem {
    float a[2][BM * LDK];
    float b[2][BM * LDK];  // NT: [BN][LDK]; NN: [BK][LDN] (smaller, fits)
};

template <bool TRANS_B>
__global__ void __launch_bounds__(256, 2)
gemm_f32_mfma_kernel(gemm_args g)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    f32_smem &sm = *reinterpret_cast<f32_smem *>(smem_raw);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5, l31 = lane & 31;

    uint32_t tm, tn, batch_u;
    batched_tile_coords(g.tiles_m, g.tiles_n, g.group_m, tm, tn, batch_u);
    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
    const int64_t batch = batch_u;
    const float *__restrict__ A = static_cast<const float *>(g.a) + batch * g.stride_a;
    const float *__restrict__ B = static_cast<const float *>(g.b) + batch * g.stride_b;
    // split-K (round 4): K slice z of g.split_k covers K-tiles [kt0, kt0 + nk) and writes partial slab z (f32, folded in slice order by
    // gemm_splitk.hip) -- 1024^3 has 64 tiles for 256 CUs: 74.5 us unsplit, the bound of a 128 x 128 x 1024 tile on one CU's f32 matrix pipe
    const int nk_total = (int)(g.k / BK);
    const int z = (g.split_k > 1) ? (int)blockIdx.z : 0;
    const int per = (nk_total + (int)max(g.split_k, 1u) - 1) / (int)max(g.split_k, 1u);
    const int kt0 = z * per;
    float *__restrict__ C = static_cast<float *>(g.c) + batch * g.stride_c + (int64_t)z * g.split_c_stride;

    // ---- global -> register staging map (4 x float4 per operand per thread) -----------------
    const float *pa[4];
    const float *pb[4];
    int wa[4], wb[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int lin = tid + e * 256;
        {
            const int row = lin >> 3, c4 = lin & 7;
            const int64_t m = min(m0 + row, g.m - 1);
            pa[e] = A + m * g.lda + c4 * 4 + (int64_t)kt0 * BK;
            wa[e] = row * LDK + c4 * 4;
        }
        if (TRANS_B) {
            const int row = lin >> 3, c4 = lin & 7;
            const int64_t n = min(n0 + row, g.n - 1);
            pb[e] = B + n * g.ldb + c4 * 4 + (int64_t)kt0 * BK;
            wb[e] = row * LDK + c4 * 4;
        } else {
            const int krow = lin >> 5, c4 = lin & 31;
            const int64_t n = min(n0 + c4 * 4, g.n - 4);
            pb[e] = B + ((int64_t)krow + (int64_t)kt0 * BK) * g.ldb + n;
            wb[e] = krow * LDN + c4 * 4;
        }
    }
    const int64_t b_step = TRANS_B ? (int64_t)BK : (int64_t)BK * g.ldb;

    // ---- LDS fragment read offsets (floats) ---------------------------------------------------
    int ra[2], rb[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        ra[t] = (wm * 64 + t * 32 + l31) * LDK + 4 * h;
        rb[t] = TRANS_B ? (wn * 64 + t * 32 + l31) * LDK + 4 * h : (4 * h) * LDN + wn * 64 + t * 32 + l31;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = min(per, nk_total - kt0);              // (the launcher leaves no slice empty)
    f32x4 ga[4], gb[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        ga[e] = *reinterpret_cast<const f32x4 *>(pa[e]);
        gb[e] = *reinterpret_cast<const f32x4 *>(pb[e]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        *reinterpret_cast<f32x4 *>(&sm.a[0][wa[e]]) = ga[e];
        *reinterpret_cast<f32x4 *>(&sm.b[0][wb[e]]) = gb[e];
    }
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = (kt + 1) < nk;
        if (more) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ga[e] = *reinterpret_cast<const f32x4 *>(pa[e] + (int64_t)(kt + 1) * BK);
                gb[e] = *reinterpret_cast<const f32x4 *>(pb[e] + (int64_t)(kt + 1) * b_step);
            }
        }
        const float *__restrict__ sa = sm.a[cur];
        const float *__restrict__ sb = sm.b[cur];
#pragma unroll
        for (int t = 0; t < BK / 8; ++t) {
            f32x4 af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const f32x4 *>(sa + ra[i] + 8 * t);
            if (TRANS_B) {
#pragma unroll
                for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const f32x4 *>(sb + rb[j] + 8 * t);
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) bf[j][q] = sb[rb[j] + (8 * t + q) * LDN];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[j][q], af[i][q], acc[i][j], 0, 0, 0);
        }
        if (more) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                *reinterpret_cast<f32x4 *>(&sm.a[cur ^ 1][wa[e]]) = ga[e];
                *reinterpret_cast<f32x4 *>(&sm.b[cur ^ 1][wb[e]]) = gb[e];
            }
        }
        __syncthreads();
    }

    // ---- epilogue: lane holds C[m][n_base + 0..3] in regs 4q..4q+3 --------------------------
    const bool vec_ok = ((g.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(C) & 15u) == 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int64_t m = m0 + wm * 64 + i * 32 + l31;
        if (m >= g.m) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t n = n0 + wn * 64 + j * 32 + 8 * q + 4 * h;
                float *dst = C + m * g.ldc + n;
                if (vec_ok && n + 3 < g.n) {
                    f32x4 v = {acc[i][j][4 * q + 0], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                    *reinterpret_cast<f32x4 *>(dst) = v;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (n + r < g.n) dst[r] = acc[i][j][4 * q + r];
                }
            }
        }
    }
}

}  // namespace

namespace mi355 {

bool gemm_f32_mfma_supports(const mi355_gemm_desc &d, const void *a, const void *b, const void *c)
{
    (void)c;
    if (d.dtype_ab != MI355_DTYPE_F32 || d.dtype_c != MI355_DTYPE_F32) return false;
    if (d.trans_a) return false;
    if (d.k < BK || d.k % BK != 0) return false;
    if (d.m < 1 || d.n < 1) return false;
    if ((d.lda & 3) || (d.ldb & 3)) return false;
    if ((d.stride_a & 3) || (d.stride_b & 3)) return false;
    if ((reinterpret_cast<uintptr_t>(a) & 15u) || (reinterpret_cast<uintptr_t>(b) & 15u)) return false;
    if (!d.trans_b && ((d.n & 3) || d.n < 4)) return false;
    if (d.batch > 65535) return false;
    const int64_t tiles = ((d.m + BM - 1) / BM) * ((d.n + BN - 1) / BN);
    if (tiles * std::max<int64_t>(d.batch, 1) > 0x7FFFFFFF) return false;   // 32-bit (batch, tile) sequence for the XCD remap
    return true;
}

int32_t launch_gemm_f32_mfma(mi355_ctx *ctx, hipStream_t s, const mi355_gemm_desc &d, const void *a, const void *b,
                             void *c)
{
    if (!gemm_f32_mfma_supports(d, a, b, c))
        return fail(ctx, MI355_E_UNSUPPORTED, "f32 MFMA GEMM: shape/layout not supported by this kernel");
    gemm_args g{};
    g.a = a; g.b = b; g.c = c;
    g.m = d.m; g.n = d.n; g.k = d.k;
    g.lda = d.lda; g.ldb = d.ldb; g.ldc = d.ldc;
    g.stride_a = d.stride_a; g.stride_b = d.stride_b; g.stride_c = d.stride_c;
    g.tiles_m = (uint32_t)((d.m + BM - 1) / BM);
    g.tiles_n = (uint32_t)((d.n + BN - 1) / BN);
    g.group_m = 8;
    // Split-K when the tiles cannot fill the chip and K is long enough to cut: the slabs are f32 like the result, so the fold adds
    // (splits + 1) x M x N x 4 bytes of traffic -- kept below the operand bytes.  1024^3: 64 tiles x 4 slices.
    const int64_t tiles = (int64_t)g.tiles_m * g.tiles_n * std::max<int64_t>(d.batch, 1), cus = ctx->props.num_streaming_multiprocessors;
    const int64_t nk_all = d.k / BK;
    int64_t splits = 1;
    if (tiles * 2 <= cus && nk_all >= 16) {
        splits = std::min<int64_t>({cus / tiles, nk_all / 8, 16});
        // worth it while the time the cut saves (a 128 x 128 x K tile on one CU's f32 matrix pipe: 614 GFLOP/s) is at least twice what the
        // slabs cost (written once, read once, ~3 TB/s) plus the fold's launch
        const double tile_us = 2.0 * BM * BN * (double)d.k / 614e3, slab_us = (double)(d.batch * d.m * d.n * 4) / 3e6;
        while (splits > 1 && (splits + 1) * slab_us + 3.0 > 0.5 * tile_us * (1.0 - 1.0 / (double)splits)) --splits;
        if (splits > 1) { const int64_t per = (nk_all + splits - 1) / splits; splits = (nk_all + per - 1) / per; }   // no empty slices
    }
    float *ws = nullptr;
    if (splits > 1 && splitk_scratch(ctx, s, (size_t)(splits * d.batch * d.m * d.n) * sizeof(float), &ws) == MI355_OK) {
        g.c = ws; g.ldc = d.n; g.stride_c = d.m * d.n;
        g.split_k = (uint32_t)splits; g.split_c_stride = d.batch * d.m * d.n;
    } else splits = 1;
    const dim3 grid(g.tiles_m * g.tiles_n, (uint32_t)d.batch, (uint32_t)splits);
    const size_t lds = sizeof(f32_smem);
    // the dynamic-LDS attribute is per device: remember it per context, not in a process-wide static
    const void *fn = d.trans_b ? reinterpret_cast<const void *>(gemm_f32_mfma_kernel<true>)
                               : reinterpret_cast<const void *>(gemm_f32_mfma_kernel<false>);
    lds_opt_in(ctx, fn, (int)lds);
    if (d.trans_b) hipLaunchKernelGGL(gemm_f32_mfma_kernel<true>, grid, dim3(256), lds, s, g);
    else hipLaunchKernelGGL(gemm_f32_mfma_kernel<false>, grid, dim3(256), lds, s, g);
    check_launch(ctx, "mi355_gemm(f32 mfma)");
    if (splits > 1) {
        launch_splitk_fold(s, ws, (uint32_t)splits, d.batch * d.m * d.n, d.batch, d.m, d.n, c, MI355_DTYPE_F32, d.ldc, d.stride_c);
        check_launch(ctx, "mi355_gemm(f32 split-K fold)");
    }
    return MI355_OK;
}

}  // namespace mi355
