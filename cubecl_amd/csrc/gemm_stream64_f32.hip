// gemm_stream64_f32.hip -- f32 GEMM with 5 ... 64 rows (or columns; rule::F32_STREAM_MIN_ROWS in gemm.cpp -- it runs from one row up, the FMA kernel
// is faster below five), both operands K-contiguous: C[m][n] = sum_k A[m][k] * B[n][k].
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
#include <type_traits>

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

template <int N, typename F, int I = 0> __device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, F, I + 1>(static_cast<F &&>(f));
    }
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

    // piece T of a stage (the streamed blocks' first), by itself
    auto issue_one = [&](int stage, int it, auto tt) {
        constexpr int T = decltype(tt)::value, B = T / 4, Q = T % 4;
        const int64_t koff = (int64_t)(wave + it * WAVES) * KB;
        const uint32_t dst = lds_addr_of(ring + stage * STAGE_BYTES) + B * BLOCK_BYTES + Q * 1024;
        if constexpr (B < NBW) glds16<NT>(big + koff, voff_big[B][Q], dst);
        else glds16<false>(small_ + koff, voff_small[B - NBW][Q], dst);
    };
    // one stage: fragments out of LDS (lane: row j, quad gq of each of the four 16-k chunks), then 16 MB NBW MFMAs with the slot's refill
    // dealt out among them -- a DMA piece costs its issuing wave ~100 cycles, and in a block of twelve in front of the MFMAs the matrix
    // pipe sat idle for all of them (profiles/r05_stream64_f32_rows_form.txt)
    auto consume = [&](int stage, int refill_it, auto refill_c) {
        constexpr bool refill = decltype(refill_c)::value;
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
        __builtin_amdgcn_sched_barrier(0);
        static_for<16>([&](auto gg) {
            constexpr int G = decltype(gg)::value, C = G / 4, E = G % 4;
#pragma unroll
            for (int a = 0; a < MB; ++a)
#pragma unroll
                for (int b = 0; b < NBW; ++b)
                    acc[a][b][E & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fs[a][C][E], fw[b][C][E], acc[a][b][E & 1], 0, 0, 0);
            if constexpr (refill) {
                static_for<(G + 1) * PIECES / 16 - G * PIECES / 16>([&](auto tt) {
                    issue_one(stage, refill_it, std::integral_constant<int, G * PIECES / 16 + decltype(tt)::value>{});
                });
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    // prologue: S stages in flight (a wave with fewer K-blocks than stages issues what it has)
    const int pre = count < S ? count : S;
    for (int s = 0; s < pre; ++s) issue(s, s);
    int it = 0, stage = 0;
    for (; it + S < count; ++it) {                  // steady state: S - 1 younger stages stay in flight
        wait_vmcnt<(S - 1) * PIECES>();
        consume(stage, it + S, std::true_type{});
        stage = stage + 1 == S ? 0 : stage + 1;
    }
    for (; it < count; ++it) {                                                    // the last S stages: nothing left to issue
        wait_vmcnt<0>();
        consume(stage, 0, std::false_type{});
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

// 49 ... 64 small rows (MB = 4 blocks; measured with three too: never ahead): the form above gives every wave its own copy of all MB small blocks, which leaves the ring
// one stage of look-ahead (16 KiB of the streamed operand in flight per CU: 64 x 8192 x 8192 122 us).  Here the waves split the SMALL
// rows instead of K: wave w owns small block w and its 16 x 32 piece of the output, all MB waves walk the same K-blocks and SHARE the
// two streamed blocks of a stage, which they fetch together (wave w issues pieces w, w + MB, ...) -- one s_barrier per K-block ("my
// pieces of stage i have landed" / "everyone is done reading stage i - 1", whose slot is then refilled), six stages deep: 40 KiB of
// the streamed operand in flight.  No K split, so no final fold.  The bound is the f32 matrix core: 32 MFMAs x 32 cycles per wave
// and K-block of 64 -- 131 k cycles = 62 us at 64 x 8192 x 8192 (measured: 97.5, against 106.1 on the K-split form; the refill's DMA pieces are
// dealt out among the MFMAs in both forms: issued as a block in front of them they cost 64 rows 15 us).
template <int MB, int S, bool NT>
__global__ __launch_bounds__(MB * 64) void gemm_stream64_f32_rows_kernel(rows_args g)
{
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    constexpr int NBW = 2, STAGE_BYTES = (NBW + MB) * BLOCK_BYTES, WSHARE = (NBW * 4 + MB - 1) / MB, PIECES = 4 + WSHARE;
    static_assert((S - 1) * PIECES <= 63 && S * STAGE_BYTES <= 160 * 1024, "pieces in flight fit vmcnt, the ring fits the CU's LDS");
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, gq = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * (16 * NBW);
    const float *small_ = static_cast<const float *>(g.small_) + (int64_t)blockIdx.y * g.stride_small;
    const float *big = static_cast<const float *>(g.big) + (int64_t)blockIdx.y * g.stride_big + row0 * g.ld_big;

    // this wave's pieces of a stage: four of its own small block, WSHARE of the shared streamed blocks (piece p = wave + t MB, modulo 8:
    // with three waves one piece is fetched twice -- same bytes, same place)
    uint32_t voff_small[4], voff_big[WSHARE], dst_big[WSHARE];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = 4 * q + gq;
        voff_small[q] = (uint32_t)(((int64_t)min(16 * wave + r, g.small_rows - 1) * g.ld_small + (j ^ r) * 4) * 4);
    }
#pragma unroll
    for (int t = 0; t < WSHARE; ++t) {
        const int p = (wave + t * MB) & 7, b = p >> 2, q = p & 3, r = 4 * q + gq;
        const int64_t row = min((int64_t)(16 * b + r), (int64_t)g.big_rows - 1 - row0);
        voff_big[t] = (uint32_t)((row * g.ld_big + (j ^ r) * 4) * 4);
        dst_big[t] = (uint32_t)(b * BLOCK_BYTES + q * 1024);
    }
    const int count = g.k / KB;
    auto issue = [&](int stage, int it) {
        const int64_t koff = (int64_t)it * KB;
        const uint32_t dst = lds_addr_of(smem + stage * STAGE_BYTES);
#pragma unroll
        for (int t = 0; t < WSHARE; ++t) glds16<NT>(big + koff, voff_big[t], dst + dst_big[t]);
#pragma unroll
        for (int q = 0; q < 4; ++q) glds16<false>(small_ + koff, voff_small[q], dst + (NBW + wave) * BLOCK_BYTES + q * 1024);
    };
    f32x4 acc[NBW][2];
#pragma unroll
    for (int b = 0; b < NBW; ++b) acc[b][0] = acc[b][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto issue_one = [&](int stage, int it, auto tt) {               // piece T of this wave's share of a stage
        constexpr int T = decltype(tt)::value;
        const int64_t koff = (int64_t)it * KB;
        const uint32_t dst = lds_addr_of(smem + stage * STAGE_BYTES);
        if constexpr (T < WSHARE) glds16<NT>(big + koff, voff_big[T], dst + dst_big[T]);
        else glds16<false>(small_ + koff, voff_small[T - WSHARE], dst + (NBW + wave) * BLOCK_BYTES + (T - WSHARE) * 1024);
    };
    // (refill_c: the slot `fill` takes K-block `refill_it`, its pieces dealt out among this stage's MFMAs)
    auto compute = [&](int stage, int fill, int refill_it, auto refill_c) {
        constexpr bool refill = decltype(refill_c)::value;
        const char *st = smem + stage * STAGE_BYTES + j * ROW_BYTES;
        f32x4 fw[NBW][4], fs[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int phys = ((4 * c + gq) ^ j) * 16;
#pragma unroll
            for (int b = 0; b < NBW; ++b) fw[b][c] = *reinterpret_cast<const f32x4 *>(st + b * BLOCK_BYTES + phys);
            fs[c] = *reinterpret_cast<const f32x4 *>(st + (NBW + wave) * BLOCK_BYTES + phys);
        }
        // all twelve reads are in flight before the first MFMA (left alone the compiler re-uses sixteen registers and waits for every chunk's reads
        // behind the previous chunk's MFMAs: four exposed LDS latencies per K-block); no explicit lgkmcnt wait: every fragment feeds an MFMA below,
        // so the reads of this stage are complete before the wave can reach the next barrier
        __builtin_amdgcn_sched_barrier(0);
        static_for<16>([&](auto gg) {
            constexpr int G = decltype(gg)::value, C = G / 4, E = G % 4;
#pragma unroll
            for (int b = 0; b < NBW; ++b) acc[b][E & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fs[C][E], fw[b][C][E], acc[b][E & 1], 0, 0, 0);
            if constexpr (refill) {
                static_for<(G + 1) * PIECES / 16 - G * PIECES / 16>([&](auto tt) {
                    issue_one(fill, refill_it, std::integral_constant<int, G * PIECES / 16 + decltype(tt)::value>{});
                });
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    const int pre = count < S - 1 ? count : S - 1;
    for (int s = 0; s < pre; ++s) issue(s, s);
    int it = 0, stage = 0, fill = S - 1;                            // `fill`: the slot of stage it - 1 (free behind the barrier)
    for (; it + S - 1 < count; ++it) {
        wait_vmcnt<(S - 2) * PIECES>();                              // my pieces of stage `it` have landed; S - 2 younger stages may fly
        __builtin_amdgcn_s_barrier();
        compute(stage, fill, it + S - 1, std::true_type{});
        fill = stage;
        stage = stage + 1 == S ? 0 : stage + 1;
    }
    for (; it < count; ++it) {
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        compute(stage, 0, 0, std::false_type{});
        stage = stage + 1 == S ? 0 : stage + 1;
    }
#pragma unroll
    for (int b = 0; b < NBW; ++b) {
        const f32x4 sum = acc[b][0] + acc[b][1];
        const int64_t n = row0 + 16 * b + j;
        if (n >= g.big_rows) continue;
        float *out = g.out + (int64_t)blockIdx.y * g.stride_out + n * g.out_stride_big;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = 16 * wave + 4 * gq + r;
            if (m < g.small_rows) out[(int64_t)m * g.out_stride_small] = sum[r];
        }
    }
}

template <int MB, int S>
void launch_rows_form(mi355_ctx *ctx, hipStream_t s, const rows_args &g, uint32_t batch)
{
    constexpr int LDS = S * (2 + MB) * BLOCK_BYTES;
    const uint32_t nblocks = (uint32_t)((g.big_rows + 31) / 32);
    if ((int64_t)g.big_rows * g.k * 4 * batch > (192ll << 20)) {
        lds_opt_in(ctx, reinterpret_cast<const void *>(gemm_stream64_f32_rows_kernel<MB, S, true>), LDS);
        hipLaunchKernelGGL((gemm_stream64_f32_rows_kernel<MB, S, true>), dim3(nblocks, batch), dim3(MB * 64), LDS, s, g);
    } else {
        lds_opt_in(ctx, reinterpret_cast<const void *>(gemm_stream64_f32_rows_kernel<MB, S, false>), LDS);
        hipLaunchKernelGGL((gemm_stream64_f32_rows_kernel<MB, S, false>), dim3(nblocks, batch), dim3(MB * 64), LDS, s, g);
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

// 49 ... 64 small rows: the rows-split form (gemm_stream64_f32_rows_kernel) where its 32-row workgroups make one to two rounds of the chip.
// Cold, us, rows-split / K-split (profiles/r05_stream64_f32_rows_form.txt): 64 x 8192 x 8192 97.5 / 106.1, 8192 x 64 x 8192 85.5 / 91.1,
// 64 x 14336 x 4096 85.5 / 92.9, 56 x 16384 x 2048 46.1 / 56.0; not taken: 64 x 4096 x 4096 (half a round) 41.6 / 27.4, 64 x 28672 x 4096 (3.5 rounds)
// 167.9 / 163.0, and 33 ... 48 rows anywhere (48 x 8192 x 8192 86.7 / 81.6).  MI355_S64F_ROWS=0 / 1 (dev) forces the choice for 49 ... 64 rows.
bool stream64_f32_rows_form(const mi355_gemm_desc &d, int cus)
{
    const int64_t small_rows = std::min(d.m, d.n), wgs = (std::max(d.m, d.n) + 31) / 32 * d.batch;
    if (small_rows <= 48) return false;
    static const int forced = [] { const char *e = getenv("MI355_S64F_ROWS"); return e ? atoi(e) : -1; }();
    if (forced == 0 || forced == 1) return forced == 1;
    return wgs >= cus && wgs <= 2 * cus;
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
    else if (stream64_f32_rows_form(d, cus)) launch_rows_form<4, 6>(ctx, s, g, batch);
    else launch_form<4, 1, 2>(ctx, s, g, batch);
    check_launch(ctx, "mi355_gemm(stream64, f32)");
    return MI355_OK;
}

}  // namespace mi355
