// gemm_stream64_f32.hip -- f32 GEMM with 9 ... 64 rows (or columns), both operands K-contiguous: C[m][n] = sum_k A[m][k] * B[n][k].
// The f32 form of MI355_GEMM_ALGO_STREAM64 (round 5; until then these shapes ran on the 128x128 f32 tile: 16 x 8192 x 8192 150 us).
//
// Roofline: HBM up to 32 rows (the large operand, 256 MiB at 8192 x 8192, is read once), the f32 matrix core from there
// (v_mfma_f32_16x16x4_f32: 256 FLOP / clk / CU -- 64 x 8192 x 8192 needs 55 us of it).
//
//   * the instruction's operand layout is lane -> (row = lane % 16, k = lane / 16): consecutive lanes hold different rows, so
//     fragment-shaped global loads would touch 16 lines per instruction, 64 bytes of each.  Both operands go through LDS instead:
//     one LDS-DMA piece (`global_load_lds_dwordx4`, 1 KiB) fetches 256 contiguous bytes = 64 k-values of each of four rows -- whole
//     lines -- into a [16 rows][256 B] block whose 16-byte pieces are XOR-swizzled by the row (the swizzle is applied on the global
//     side: lane i of a piece fetches logical piece (i % 16) ^ row); `ds_read_b128` then hands each lane 4 consecutive k-values of
//     its row, conflict-free in all four service groups of the instruction, and those feed four MFMAs (the k index inside an MFMA
//     is whatever the two operands agree on).
//   * no barrier in the K loop: a workgroup is four waves that own the SAME 16 NBW streamed rows and split K among themselves in
//     blocks of 64 (wave w takes K-blocks w, w + 4, ...: together they read 1 KiB runs of each row); every wave fills and drains its
//     own ring of S stages and only waits for its own pieces (`s_waitcnt vmcnt`).  The four partial blocks meet once, at the end,
//     through LDS, added in wave order: deterministic.
//   * the small operand (<= 64 rows x K: L2-resident) takes the same path, once per wave: with NBW = 2 its L2 -> LDS traffic is
//     half the streamed operand's HBM traffic.
//
// Either operand may be the small one (few columns: the roles swap, the output block is stored transposed).
// Accumulation: f32, per wave K-blocks ascending with the even and odd k of each quad in two accumulators, then the waves in order --
// within the parity tolerance of tests/test_gpu_gemm.py, not bit-identical to the tile kernels (different association).
#include <algorithm>
#include <cstdlib>

#include "gemm_common.hpp"

using namespace mi355;

namespace {

constexpr int WAVES = 4;
constexpr int KB = 64;                       // k-values of one K-block
constexpr int ROW_BYTES = KB * 4;            // 256: one row of a block = every LDS bank once
constexpr int BLOCK_BYTES = 16 * ROW_BYTES;  // 4 KiB: 16 rows of one operand, four DMA pieces

struct rows_args {
    const void *small_;      // [small_rows][K]
    const void *big;         // [big_rows][K]
    float *out;
    int64_t ld_small, ld_big;                       // elements
    int64_t out_stride_small, out_stride_big;       // elements between consecutive small / big indices of the output
    int64_t stride_small, stride_big, stride_out;   // batch strides, elements
    int32_t small_rows, big_rows, k;
};

template <bool NT>
__device__ __forceinline__ void glds16(const void *ubase_in, uint32_t voff, uint32_t lds_in)
{
    const uint64_t u = reinterpret_cast<uint64_t>(ubase_in);
    const uint64_t us = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32)) << 32) |
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u);
    const void *ubase = reinterpret_cast<const void *>(us);
    const uint32_t lds_byte_addr = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_in);
    // s_nop 4: five wait states between a VALU write of an SGPR (the readfirstlanes) and a VMEM read of it (tools/hazard_scan.py)
    if constexpr (NT)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(voff), "s"(ubase), "s"(lds_byte_addr) : "memory");
    else
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(ubase), "s"(lds_byte_addr) : "memory");
}
__device__ __forceinline__ uint32_t lds_addr_of(const void *p)
{
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char *)p;
}

template <int N> __device__ __forceinline__ void wait_vmcnt()
{
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// MB: 16-row blocks of the small operand; NBW: 16-row blocks of the streamed operand per workgroup; S: ring stages per wave;
// NT: the streamed operand cannot stay in the Infinity Cache anyway -- non-temporal pieces (as gemm_stream64.hip)
template <int MB, int NBW, int S, bool NT>
__global__ __launch_bounds__(WAVES * 64) void gemm_stream64_f32_kernel(rows_args g)
{
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    constexpr int BLOCKS = NBW + MB, STAGE_BYTES = BLOCKS * BLOCK_BYTES, PIECES = 4 * BLOCKS;   // DMA pieces of one stage
    static_assert((S - 1) * PIECES <= 63, "a wave's pieces in flight must fit vmcnt");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, gq = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * (16 * NBW);
    const float *small_ = static_cast<const float *>(g.small_) + (int64_t)blockIdx.y * g.stride_small;
    const float *big = static_cast<const float *>(g.big) + (int64_t)blockIdx.y * g.stride_big + row0 * g.ld_big;
    char *ring = smem + wave * (S * STAGE_BYTES);

    // piece q of a block: lane i fetches, for row 4q + i / 16, the 16 bytes that belong at physical position i % 16 of the row's 256
    // (rows past the end re-read the last one: loaded, multiplied, never stored)
    uint32_t voff_big[NBW][4], voff_small[MB][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = 4 * q + gq, logical = j ^ r;
#pragma unroll
        for (int b = 0; b < NBW; ++b) {
            const int64_t row = min((int64_t)(16 * b + r), (int64_t)g.big_rows - 1 - row0);
            voff_big[b][q] = (uint32_t)((row * g.ld_big + logical * 4) * 4);
        }
#pragma unroll
        for (int b = 0; b < MB; ++b) {
            const int64_t row = min(16 * b + r, g.small_rows - 1);
            voff_small[b][q] = (uint32_t)((row * g.ld_small + logical * 4) * 4);
        }
    }
    const int nkb = g.k / KB;
    const int count = wave < nkb ? (nkb - wave + WAVES - 1) / WAVES : 0;          // this wave's K-blocks: wave, wave + 4, ...

    auto issue = [&](int stage, int it) {
        const int64_t koff = (int64_t)(wave + it * WAVES) * KB;                   // elements
        const uint32_t dst = lds_addr_of(ring + stage * STAGE_BYTES);
#pragma unroll
        for (int b = 0; b < NBW; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) glds16<NT>(big + koff, voff_big[b][q], dst + b * BLOCK_BYTES + q * 1024);
#pragma unroll
        for (int b = 0; b < MB; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) glds16<false>(small_ + koff, voff_small[b][q], dst + (NBW + b) * BLOCK_BYTES + q * 1024);
    };

    f32x4 acc[MB][NBW][2];
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b = 0; b < NBW; ++b) acc[a][b][0] = acc[a][b][1] = f32x4{0.f, 0.f, 0.f, 0.f};

    // one stage: fragments out of LDS (lane: row j, quad gq of each of the four 16-k chunks), the slot refilled, 16 MB NBW MFMAs
    auto consume = [&](int stage, int refill_it, bool refill) {
        const char *st = ring + stage * STAGE_BYTES + j * ROW_BYTES;
        f32x4 fw[NBW][4], fs[MB][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int phys = ((4 * c + gq) ^ j) * 16;
#pragma unroll
            for (int b = 0; b < NBW; ++b) fw[b][c] = *reinterpret_cast<const f32x4 *>(st + b * BLOCK_BYTES + phys);
#pragma unroll
            for (int b = 0; b < MB; ++b) fs[b][c] = *reinterpret_cast<const f32x4 *>(st + (NBW + b) * BLOCK_BYTES + phys);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                        // the slot is read out: it may be overwritten
        if (refill) issue(stage, refill_it);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int a = 0; a < MB; ++a)
#pragma unroll
                    for (int b = 0; b < NBW; ++b)
                        acc[a][b][e & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fs[a][c][e], fw[b][c][e], acc[a][b][e & 1], 0, 0, 0);
    };

    // prologue: S stages in flight (a wave with fewer K-blocks than stages issues what it has)
    const int pre = count < S ? count : S;
    for (int s = 0; s < pre; ++s) issue(s, s);
    int it = 0, stage = 0;
    for (; it + S < count; ++it) {                  // steady state: S - 1 younger stages stay in flight
        wait_vmcnt<(S - 1) * PIECES>();
        consume(stage, it + S, true);
        stage = stage + 1 == S ? 0 : stage + 1;
    }
    for (; it < count; ++it) {                                                    // the last S stages: nothing left to issue
        wait_vmcnt<0>();
        consume(stage, 0, false);
        stage = stage + 1 == S ? 0 : stage + 1;
    }

    // ---- the four waves' partial blocks meet in LDS; wave w adds fragment f = w, w + 4, ... in wave order and stores it -------------
    __syncthreads();                                                              // every ring is drained: the memory is free
    f32x4 *red = reinterpret_cast<f32x4 *>(smem);                                 // [wave][MB x NBW][64 lanes]
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b = 0; b < NBW; ++b) red[(wave * (MB * NBW) + a * NBW + b) * 64 + lane] = acc[a][b][0] + acc[a][b][1];
    __syncthreads();
    for (int f = wave; f < MB * NBW; f += WAVES) {
        f32x4 sum = red[f * 64 + lane];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) sum += red[(w * (MB * NBW) + f) * 64 + lane];
        const int a = f / NBW, b = f % NBW;
        const int64_t n = row0 + 16 * b + j;                                      // lane: streamed row j of the block, small rows 4 gq ... + 3
        if (n >= g.big_rows) continue;
        float *out = g.out + (int64_t)blockIdx.y * g.stride_out + n * g.out_stride_big;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = 16 * a + 4 * gq + r;
            if (m < g.small_rows) out[(int64_t)m * g.out_stride_small] = sum[r];
        }
    }
}

template <int MB, int NBW, int S>
void launch_form(mi355_ctx *ctx, hipStream_t s, const rows_args &g, uint32_t batch)
{
    constexpr int LDS = WAVES * S * (NBW + MB) * BLOCK_BYTES;
    static_assert(LDS <= 160 * 1024 && LDS >= WAVES * MB * NBW * 1024, "rings fit the CU's LDS and hold the final fold");
    const uint32_t nblocks = (uint32_t)((g.big_rows + 16 * NBW - 1) / (16 * NBW));
    // streamed bytes of the whole launch against what the 256 MiB Infinity Cache could still hold next to everything else
    if ((int64_t)g.big_rows * g.k * 4 * batch > (192ll << 20)) {
        lds_opt_in(ctx, reinterpret_cast<const void *>(gemm_stream64_f32_kernel<MB, NBW, S, true>), LDS);
        hipLaunchKernelGGL((gemm_stream64_f32_kernel<MB, NBW, S, true>), dim3(nblocks, batch), dim3(WAVES * 64), LDS, s, g);
    } else {
        lds_opt_in(ctx, reinterpret_cast<const void *>(gemm_stream64_f32_kernel<MB, NBW, S, false>), LDS);
        hipLaunchKernelGGL((gemm_stream64_f32_kernel<MB, NBW, S, false>), dim3(nblocks, batch), dim3(WAVES * 64), LDS, s, g);
    }
}

}  // namespace

namespace mi355 {

// A [M][K], B [N][K] K-contiguous f32, f32 C, min(M, N) <= 64, K a multiple of 64, 16-byte aligned rows.
bool gemm_stream64_f32_supports(const mi355_gemm_desc &d, const void *a, const void *b, const void *c)
{
    (void)c;
    if (d.dtype_ab != MI355_DTYPE_F32 || d.dtype_c != MI355_DTYPE_F32) return false;
    if (d.trans_a || !d.trans_b) return false;
    if (d.m <= 0 || d.n <= 0 || d.k < KB || (d.k & (KB - 1)) || d.k > 0x7FFFFFC0) return false;
    if (std::min(d.m, d.n) > 64) return false;
    if (d.m > 0x7FFFFFFF || d.n > 0x7FFFFFFF || d.batch < 1 || d.batch > 65535) return false;
    if ((d.lda & 3) || (d.ldb & 3) || (d.stride_a & 3) || (d.stride_b & 3)) return false;
    if ((reinterpret_cast<uintptr_t>(a) & 15u) || (reinterpret_cast<uintptr_t>(b) & 15u)) return false;
    if ((int64_t)64 * std::max(d.lda, d.ldb) * 4 >= (1ll << 32)) return false;     // per-lane DMA offsets are 32-bit
    return true;
}

// 16-row blocks of the streamed operand per workgroup.  A workgroup's rings fill the CU's LDS, so workgroups run in rounds of one per
// CU and a round of the two-block form takes twice as long: two blocks (half the small operand's L2 -> LDS traffic) unless one block
// needs fewer round-units.  Cold, us, one / two blocks (profiles/r05_stream64_f32_blocks.txt): 16 x 8192 x 8192 (512 / 256 workgroups)
// 62.4 / 54.6, 16 x 6144 x 6144 (384 / 192) 40.6 / 34.0, 16 x 5120 x 8192 (320 / 160) 52.0 / 40.0, 32 x 8192 x 8192 73.7 / 59.1; the other way:
// 16 x 4096 x 4096 (256 / 128) 17.1 / 21.4, 16 x 28672 x 4096 (7 rounds / 3.5) 96.7 / 112.4.
// A pure function of the descriptor and the CU count (MI355_S64F_NBW=1 / 2, dev, forces it).
int stream64_f32_blocks(const mi355_gemm_desc &d, int cus)
{
    const int64_t small_rows = std::min(d.m, d.n), big_rows = std::max(d.m, d.n);
    if (small_rows > 32) return 1;                                                  // (three or four small blocks: the LDS holds one streamed block per stage)
    static const int forced = [] { const char *e = getenv("MI355_S64F_NBW"); return e ? atoi(e) : 0; }();
    if (forced == 1 || forced == 2) return forced;
    const int64_t units1 = ((big_rows + 15) / 16 * d.batch + cus - 1) / cus, units2 = 2 * (((big_rows + 31) / 32 * d.batch + cus - 1) / cus);
    return units2 <= units1 ? 2 : 1;
}

int32_t launch_gemm_stream64_f32(mi355_ctx *ctx, hipStream_t s, const mi355_gemm_desc &d, const void *a, const void *b, void *c)
{
    if (!gemm_stream64_f32_supports(d, a, b, c)) return fail(ctx, MI355_E_UNSUPPORTED, "mi355_gemm: the f32 streaming kernel does not take this descriptor");
    rows_args g{};
    const bool a_small = d.m <= d.n;
    g.small_ = a_small ? a : b;
    g.big = a_small ? b : a;
    g.out = static_cast<float *>(c);
    g.ld_small = a_small ? d.lda : d.ldb;
    g.ld_big = a_small ? d.ldb : d.lda;
    g.out_stride_small = a_small ? d.ldc : 1;
    g.out_stride_big = a_small ? 1 : d.ldc;
    g.stride_small = a_small ? d.stride_a : d.stride_b;
    g.stride_big = a_small ? d.stride_b : d.stride_a;
    g.stride_out = d.stride_c;
    g.small_rows = (int32_t)(a_small ? d.m : d.n);
    g.big_rows = (int32_t)(a_small ? d.n : d.m);
    g.k = (int32_t)d.k;
    const uint32_t batch = (uint32_t)d.batch;
    const int cus = ctx->props.num_streaming_multiprocessors > 0 ? ctx->props.num_streaming_multiprocessors : 256;
    const int nbw = stream64_f32_blocks(d, cus);
    if (g.small_rows <= 16) {
        if (nbw == 2) launch_form<1, 2, 3>(ctx, s, g, batch);
        else launch_form<1, 1, 4>(ctx, s, g, batch);
    } else if (g.small_rows <= 32) {
        if (nbw == 2) launch_form<2, 2, 2>(ctx, s, g, batch);
        else launch_form<2, 1, 3>(ctx, s, g, batch);
    } else if (g.small_rows <= 48) launch_form<3, 1, 2>(ctx, s, g, batch);
    else launch_form<4, 1, 2>(ctx, s, g, batch);
    check_launch(ctx, "mi355_gemm(stream64, f32)");
    return MI355_OK;
}

}  // namespace mi355
