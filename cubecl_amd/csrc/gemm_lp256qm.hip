// gemm_lp256qm.hip -- the persistent 256 x 256 kernel with dripped C stores (gemm_lp256q.hip) on v_mfma_f32_16x16x32 with the
// stationary second source operand (gemm_lp256m16.hip).  Round 6; config C5 (batched 2048^3 bf16, 16-bit C).  Read those two
// headers first: ring, LDS image, hand-over, vmcnt discipline and rasterisation are theirs.
//
// Roofline: MFMA bf16 / f16, 2.5 PFLOP/s dense.  Why: on uniform operands the chip is power-limited; the narrow MFMA shape holds
// 1.82-1.86 GHz where 32x32x16 holds 1.52-1.54 (C5, same box), but the one-tile-per-workgroup form of the narrow kernel runs C5 at
// 0.65 of the matrix rate at its clock -- prologue and epilogue of a 32-K-tile tile are not overlapped -- where the persistent
// 32x32x16 kernel runs at 0.79 of its (lower) clock (profiles/r06_c5_baseline.txt).  This kernel is the persistent form at the
// narrow shape's clock.
//
// What differs from gemm_lp256q.hip:
//   * K loop: gemm_lp256m16.hip's -- a wave's 128 x 128 block is 8 x 8 accumulators of 16 x 16 (256 AGPRs), a K-tile two k-steps
//     of 64 MFMAs (A fragment = srcB outer, B fragment = srcA inner).  Fragment registers: the 8 A fragments are replaced IN
//     PLACE (each is dead after its 8 MFMAs of the k-step, its successor's read is issued right there: qm_read_at), the B
//     fragments double-buffered: 96 registers, not m16's 128.  The MFMAs are inline asm with the accumulator tied ("+a"): the
//     compiler can neither move an accumulator out of its AGPR nor re-order across the pinned conversions (0 spills).
//   * The held tile needs no v_permlane swap: the ROWS OF THE B TILE ARE PERMUTED ON THEIR WAY INTO LDS.  With srcA = the B
//     fragment a lane (l15 = lane % 16, g = lane / 16) holds C[row l15][MFMA columns 4 g .. 4 g + 3] of block (i, j) -- four
//     consecutive columns of one row.  LDS row r of the B tile is filled from matrix column
//         pi(r) = 32 (J >> 1) + 8 (n >> 2) + 4 (J & 1) + (n & 3),     J = r >> 4 (column block), n = r & 15
//     (a wave-uniform row offset per DMA piece + ONE per-lane offset: sub -> 8 (sub >> 2) + (sub & 3) rows), so that the lane's
//     words of the adjacent column blocks 2 jj, 2 jj + 1 are columns 32 jj + 8 g .. + 7 of its row: one 16-byte chunk, packed
//     straight out of the accumulators.  The fragment reads, their swizzle and the MFMA order are untouched (the permutation
//     lives in the DMA's SOURCE addressing), and every output element is the same chain of 16x16x32 MFMAs as in
//     gemm_lp256m16.hip: bit-identical results.
//   * Whole-line stores (partial lines are poison, gemm_lp256q.hip): the four lanes g of a row hold 64 bytes of it per chunk
//     register, so one 2 x 2 exchange between lanes l15 ^ 1 and chunk registers jj ^ 1 (quad-permute DPP + select, 4 VALU per
//     word, 32 per 16-row block row) makes a chunk register 8 rows x 128 contiguous bytes.  24 held stores per wave and tile
//     (block rows 0-5 = 96 VGPRs), block rows 6-7 through the dead ring slot at the tile boundary, as in gemm_lp256q.hip.
//   * Per-lane DMA offsets: four registers (A / B x even / odd piece) + wave-uniform piece offsets (tiles are full; the sixteen sums are
//     loop-invariant registers); fragment addresses: x1 = x0 ^ 64.  96 fragment + 96 held registers leave ~60 for everything else.
//   * The K-tile's bookkeeping lives in the MFMA gaps (second K loop of round 6; "the K-tile's bookkeeping, one phase ahead" below).  One
//     wave per SIMD: an instruction behind an MFMA whose gap holds at most one or two others costs nothing, at the head of a K-tile it
//     costs 6.4 cycles of an idle matrix pipe (profiles/r06_qm_pad_cost.txt).  So ring positions, the issue side's tile switch and K
//     offset, read and DMA addresses are computed one phase ahead behind odd MFMAs, a DMA piece is a v_mov and the load (M0 twice per
//     unit: the instruction offset of global_load_lds moves both addresses), the dripped stores sit behind the hand-over (one wait count
//     for every tile), the next tile's coordinates are worked out behind MFMA 40 of a tile's first K-tile.  Steady K-tile: 269
//     instructions for 128 MFMAs, head 1, tail 3 (tests/test_abi_cpu.py reads the assembly); mfma_util 0.81 -> 0.87 on config C3.
//
//   * Row-major rhs (BNN = true, B stored [K][N]): the B tile's LDS image is K-major, fragments come out of ds_read_b64_tr_b16
//     (half-swapped block image: no bank conflict), and the drain exchanges with v_permlane16_swap instead of the DMA-side row
//     permutation.  Same MFMA chain per element: bit-identical to the B-transposed call on B^T (tests/test_gpu_gemm.py).
//
// Restrictions: as gemm_lp256q.hip; lda != ldb allowed; the row-major-rhs form needs 64 * ldb * 2 < 2^32 (DMA voffset range).
#include <algorithm>
#include <type_traits>

#include "gemm_common.hpp"

using namespace mi355;

namespace {

constexpr int BM = 256, BN = 256;
constexpr int ROW_BYTES = 128;                    // one K-tile row = one 128-byte line: 64 x 16-bit
constexpr int UNIT_BYTES = BM * ROW_BYTES;        // 32 KiB: one ring slot
constexpr int NSLOT = 5;
constexpr int LDS_BYTES = NSLOT * UNIT_BYTES;     // 160 KiB
constexpr int NHELD = 24;                         // dripped stores per wave and tile: 6 block rows x 2 lines x (even, odd rows)

template <int DT> struct qm;
template <> struct qm<MI355_DTYPE_BF16> {
    typedef bf16x8 frag;
    // The MFMA as inline asm with the accumulator tied to an "a" operand: with the builtin, hipcc's allocator -- 96 held + 96 fragment
    // registers beside the accumulators -- kept some accumulator blocks in VGPRs, rotated the others through a[4:7] with four
    // v_accvgpr_write in front of every second MFMA and spilled 384 registers (tools/dev/spill_map.py).  An asm operand of class "a"
    // at all 128 uses per K-tile leaves it nothing to move.  (Hazards: no MFMA reads an accumulator written less than 64 MFMAs
    // earlier; the packing reads one QM_LAG >= 2 MFMAs = 64+ cycles after its last write.)
    static __device__ __forceinline__ void mfma(frag a, frag b, f32x4 &c) { asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b)); }
    static __device__ __forceinline__ void mfma0(frag a, frag b, f32x4 &c) { asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b)); }
    static __device__ __forceinline__ uint32_t pack2(float x, float y)
    {
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        const bf16x2 v = {(__bf16)x, (__bf16)y};
        return __builtin_bit_cast(uint32_t, v);
    }
};
template <> struct qm<MI355_DTYPE_F16> {
    typedef f16x8 frag;
    static __device__ __forceinline__ void mfma(frag a, frag b, f32x4 &c) { asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b)); }
    static __device__ __forceinline__ void mfma0(frag a, frag b, f32x4 &c) { asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b)); }
    static __device__ __forceinline__ uint32_t pack2(float x, float y)
    {
        typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
        const f16x2 v = {(_Float16)x, (_Float16)y};
        return __builtin_bit_cast(uint32_t, v);
    }
};

// LDS-DMA with a wave-uniform 64-bit base in SGPRs + a constant 32-bit per-lane offset (see gemm_lp256w4.hip)
template <int IMM>
__device__ __forceinline__ void glds16_s(const void *ubase, uint32_t voff, uint32_t lds_byte_addr)
{
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(ubase), "s"(lds_byte_addr), "i"(IMM)
                 : "memory", "scc");
}
__device__ __forceinline__ uint32_t lds_addr_of(const void *p)
{
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char *)p;
}

#ifndef QM_ABL
#define QM_ABL 0          // dev ablations, timing only: 1 no dripped stores, 2 no boundary stores, 4 every dripped store of a wave into the same 8 KiB (cache-resident: the instruction without its memory traffic)
#endif
#ifndef QM_PAD
#define QM_PAD 0          // dev, timing only -- what an instruction costs by where it sits: 1: 16 scalar adds at the head of every K-tile, 2: one in each of 16 empty MFMA gaps, 4: one more in each of the 16 gaps that carry a DMA piece
#endif
#ifndef QM_NT
#define QM_NT 1           // the held tile's stores carry the streaming hint (0: plain stores, dev A/B)
#endif
#define QM_STR_(x) #x
#define QM_STR(x) QM_STR_(x)
#define QM_NT_SUFFIX_0 ""
#define QM_NT_SUFFIX_1 " nt"
#define QM_NT_CAT_(a, b) a##b
#define QM_NT_CAT(a, b) QM_NT_CAT_(a, b)
#define QM_NT_SUFFIX QM_NT_CAT(QM_NT_SUFFIX_, QM_NT)
#ifndef QM_LAG
#define QM_LAG 3          // a finished block is packed behind the MFMA this many slots after its own (no wait on the matrix pipe)
#endif
#ifndef QM_BDRIP
#define QM_BDRIP 0        // 1: the 8 boundary stores (block rows 6-7) leave one per 8 MFMAs during k-step 0 of the next tile's K-tile 0; 0: in a burst at the
                          // boundary.  Measured (C5, three interleaved pairs, profiles/r06_qm_ab.txt): 1 = 1 275-1 286, 0 = 1 288-1 303 TFLOP/s -- the stores cost
                          // the K loop what they cost wherever they sit; the burst keeps K-tile 0 clean
#endif
#ifndef QM_STORE_FORM
#define QM_STORE_FORM 1   // 1: inline asm, wave-uniform base in SGPRs + 32-bit lane offset (no address VGPRs); 0: builtin store
#endif
#define WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
template <int V> using IC = std::integral_constant<int, V>;
// Dev timing trace (-DQM_TRACE, never in the product library): shader-clock stamps of wave 0 of every workgroup for its first 8
// tiles -- {K loop entered, K loop left, boundary left} -- read back with mi355_dev_qm_trace
#ifdef QM_TRACE
__device__ unsigned long long qm_trace_buf[256 * 64];   // per workgroup: 24 stamps; [32 + 4 k + wave]: cycles in the hand-over, k = 0 vmcnt wait / 1 lgkm + barrier of the first tile, 2 / 3 of the later tiles, 4: K-tiles counted (later tiles)
#define QM_STAMP(slot) do { if (tid == 0 && blockIdx.x < 256 && qt_tile < 8) qm_trace_buf[blockIdx.x * 64 + qt_tile * 3 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#define QM_TW0() const unsigned long long qw0_ = __builtin_amdgcn_s_memtime();
#define QM_TW1() const unsigned long long qw1_ = __builtin_amdgcn_s_memtime();
#define QM_TW2() { const unsigned long long qw2_ = __builtin_amdgcn_s_memtime(); if (qt_tile == 0) { qt_w[0] += qw1_ - qw0_; qt_w[1] += qw2_ - qw1_; } else if (qt_tile < 8) { qt_w[2] += qw1_ - qw0_; qt_w[3] += qw2_ - qw1_; qt_w[4] += 1; } }
#else
#define QM_STAMP(slot)
#define QM_TW0()
#define QM_TW1()
#define QM_TW2()
#endif
// a lane-constant value the compiler must re-derive where it is used (gemm_lp256q.hip: hoisted staging addresses get spilled)
__device__ __forceinline__ uint32_t opaque(uint32_t x) { asm volatile("" : "+v"(x)); return x; }

// Which fragment read follows MFMA n of k-step ks (read id: 0..7 = B fragment j of the NEXT k-step, 8..15 = A fragment i of it; -1 none).
//   k-step 0 (the next k-step's operands are this K-tile's, landed long ago): B j behind MFMA 1 + 4 j; A i behind MFMA 8 i + 7, the last
//     MFMA that multiplies by the fragment it replaces.
//   k-step 1 (the next k-step's operands are K-tile t + 1's: nothing before the hand-over behind MFMA 15): A 0, A 1 behind MFMAs 16, 18;
//     B j behind 17 + 4 j; A i (i >= 2) behind 8 i + 7.
// DMA pieces sit behind MFMAs 3, 11 (mod 16) and, in the last quarter of k-step 1, behind 49, 53, 57, 61: no slot carries two.
// Row-major B (bnn): a B fragment is TWO transposing reads -- ids 0..7 = its k 0-3 half, 16..23 = its k 4-7 half, one MFMA slot each (two reads
// behind one MFMA cost the K loop 5 % at the clock: profiles/r06_qm_nn_ab.txt): the second half one slot behind the first, A 1 behind MFMA 20.
constexpr int qm_read_at(int ks, int n, bool bnn)
{
    if (ks == 0) {
        if ((n & 3) == 1 && n < 32) return n >> 2;
        if (bnn && (n & 3) == 2 && n < 32) return 16 + (n >> 2);
        if ((n & 7) == 7) return 8 + (n >> 3);
        return -1;
    }
    if (n == 16) return 8;
    if (n == (bnn ? 20 : 18)) return 9;
    if (n >= 17 && n < 49 && ((n - 17) & 3) == 0) return (n - 17) >> 2;
    if (bnn && n >= 18 && n < 50 && ((n - 18) & 3) == 0) return 16 + ((n - 18) >> 2);
    if (n >= 23 && (n & 7) == 7) return 8 + (n >> 3);
    return -1;
}

// BNN (round 6): B row-major [K][N], the layout TensorHandle::new_contiguous gives a rhs (crates/cubecl-std/src/tensor/handle.rs:89).
//   The matrix core wants, per lane (l15, g), the eight k-values 32 s + 8 g .. + 7 of ONE column 16 j + l15: two ds_read_b64_tr_b16 (k 0-3 and
//   4-7), each of which transposes, per 16-lane group, a [4 k][16 n] block whose sixteen 8-byte pieces the lanes address themselves (lane i: row
//   i / 4, columns 4 (i % 4) .. + 3) -- gemm_lp256w4.hip "BNN, bf16 / f16".  Image of a K-tile: 128 blocks of [4 k][32 n] = 256 contiguous bytes,
//   block (a, b) = k-rows 4 a .. + 3 x columns 32 b .. + 31 at (8 a + b) * 256, row i of it at + 64 i, as there -- EXCEPT that the two 32-byte
//   halves of a block's rows are swapped where (a >> 1) is odd.  On the 32x32x16 shape the 32 lanes of a half-wave read the two column halves
//   of one block (one whole 256-byte bank row); here they are the groups g = 0, 1: the SAME column half of two blocks 512 bytes apart -- a
//   2-way bank conflict on every read -- and with the swap (a >> 1 = 4 s + g) the two groups take opposite halves of the bank row.  The swap
//   costs nothing: LDS-DMA writes lane-linear but reads any per-lane address (two lane-offset registers, one per swap state).
//   Output: the MFMA columns keep their natural order (a DMA lane moves 16 contiguous bytes: the column permutation of the [N][K] form is not
//   available), so the held tile's packed words of column blocks 2 jj, 2 jj + 1 meet through two v_permlane16_swap per pair: lane (l15, g)
//   then holds columns 16 (2 jj + g % 2) + 8 (g / 2) .. + 7 of its row.  Same MFMA chains as gemm_lp256m16.hip would run on the transposed
//   operand: bit-identical to the [N][K] form on B^T.
// D = dripped stores per K-tile (1, 2, 4 or 8): the 24 held stores leave during K-tiles 1 .. 24 / D of the next tile
template <int DT, int D, bool BNN = false>
__global__ void __launch_bounds__(256)
gemm_lp256qm_kernel(gemm_args g)
{
    static_assert(NHELD % D == 0 && 8 % D == 0, "the held stores must split evenly over K-tiles and row blocks");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    typedef typename qm<DT>::frag frag;
    constexpr int ESZ = 2, CSZ = 2;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int g4 = lane >> 4, l15 = lane & 15;
    const int nk = (int)(g.k / 64);

    const uint32_t tiles = g.tiles_m * g.tiles_n;
    const uint32_t total = tiles * g.batch_count;
    const uint32_t lda_b = (uint32_t)(g.lda * ESZ), ldb_b = (uint32_t)(g.ldb * ESZ);    // bytes per operand row (256 rows < 2^32: supports())

    // ua / ub: this WAVE's first row of the tile's A / B panel (LDS rows 64 wave ..: pieces wave * 8 + j)
    struct tile_src { const char *ua, *ub; int64_t m0, n0, batch; };
    auto locate = [&](uint32_t L) {
        tile_src t;
        const uint32_t R = xcd_remap(L, total);
        const uint32_t bi = R / tiles, tl = R - bi * tiles;
        uint32_t tm, tn;
        tile_coords(tl, g.tiles_m, g.tiles_n, g.group_m, tm, tn);
        t.m0 = (int64_t)tm * BM; t.n0 = (int64_t)tn * BN; t.batch = bi;
        t.ua = static_cast<const char *>(g.a) + ((int64_t)bi * g.stride_a + (t.m0 + wave * 64) * g.lda) * ESZ;
        if constexpr (BNN)   // column n0 of k-row 16 wave: this wave's pieces are block rows a = 4 wave .. + 3 of every K-tile
            t.ub = static_cast<const char *>(g.b) + ((int64_t)bi * g.stride_b + t.n0 + (int64_t)(wave * 16) * g.ldb) * ESZ;
        else
            t.ub = static_cast<const char *>(g.b) + ((int64_t)bi * g.stride_b + (t.n0 + wave * 64) * g.ldb) * ESZ;
        return t;
    };
    // DMA map: a unit is 32 pieces of 1 KiB (8 LDS rows); this wave fills pieces wave * 8 + j; lane -> (LDS row r = 64 wave + 8 j + sub,
    // physical chunk c8), source chunk = c8 ^ ((r >> 1) & 7) = c8 ^ (4 (j & 1) + (sub >> 1)).  Source ROW: A r; B pi(r) =
    // 64 wave + [32 (j >> 2) + 16 (j & 1) + 4 ((j >> 1) & 1)] + [8 (sub >> 2) + (sub & 3)] -- wave-uniform piece part + lane part.
    const int sub = lane >> 3, c8 = lane & 7;
    const uint32_t voff_a0 = (uint32_t)sub * lda_b + (uint32_t)((c8 ^ (sub >> 1)) << 4);
    const uint32_t voff_a1 = (uint32_t)sub * lda_b + (uint32_t)((c8 ^ (4 + (sub >> 1))) << 4);
    // BNN: piece J of this wave = block row a = 4 wave + J / 2, blocks b = 4 (J % 2) .. + 3: lane -> block b = 4 (J % 2) + lane / 16, row
    // (lane % 16) / 4 of it, physical 16-byte chunk lane % 4 <- logical chunk (lane % 4) ^ 2 where (a >> 1) = 2 wave + J / 4 is odd
    const uint32_t voff_b0 = BNN ? (uint32_t)((lane & 15) >> 2) * ldb_b + (uint32_t)((lane >> 4) * 64 + (lane & 3) * 16)
                                 : (uint32_t)(8 * (sub >> 2) + (sub & 3)) * ldb_b + (uint32_t)((c8 ^ (sub >> 1)) << 4);
    const uint32_t voff_b1 = BNN ? (uint32_t)((lane & 15) >> 2) * ldb_b + (uint32_t)((lane >> 4) * 64 + ((lane & 3) ^ 2) * 16)
                                 : (uint32_t)(8 * (sub >> 2) + (sub & 3)) * ldb_b + (uint32_t)((c8 ^ (4 + (sub >> 1))) << 4);
    const int dst_piece = wave * 8 * 1024;

    // fragment reads (gemm_lp256m16.hip): row l15 of a 16-row block, chunk (4 s + g) ^ f, f = (l15 >> 1) & 7; k-step 1 = k-step 0 ^ 64
    const int f = (l15 >> 1) & 7;
    // BNN, B: block row 2 g (+ 8 s + u), block 4 wn (+ j / 2), row l15 / 4, half (j % 2) ^ (g % 2), 8 bytes per lane; k-step 1 = + 16 KiB
    const int ro_a = (wm * 128 + l15) * ROW_BYTES + ((g4 ^ f) << 4);
    const int ro_b = BNN ? (16 * g4 + 4 * wn) * 256 + (l15 >> 2) * 64 + (g4 & 1) * 32 + (l15 & 3) * 8 : (wn * 128 + l15) * ROW_BYTES + ((g4 ^ f) << 4);
    constexpr int KS1_B = BNN ? 16384 : 0;         // what k-step 1 adds to ro_b (the [N][K] image: ro_b ^ 64 instead)

    f32x4 acc[8][8];             // never zeroed: the first k-step of a tile accumulates into a literal zero operand
    frag fa[8], fb[2][8];        // A single-buffered IN PLACE (fragment i is re-read behind its eighth MFMA), B double-buffered
    tile_src iss;                // the tile whose K-tiles are being ISSUED (two ahead of the MFMAs)
    u32x4 P[3][2][4];            // the finished tile's block rows 0..5 packed: [rb][ii][jj] = block row 2 rb + ii, columns 32 jj + 8 g .. + 7 of row l15

    // Fragment reads.  The A fragment i (srcB) serves MFMAs 8 i .. 8 i + 7 of a k-step and nothing else, so the NEXT k-step's fragment i
    // is read into the same registers right behind MFMA 8 i + 7 (64 MFMAs before its first use): 32 registers instead of 64, which is
    // what lets the 96 held registers in.  The B fragments (srcA) are all in use until the k-step's last eight MFMAs: double-buffered.
    // Read id 0..7 = B fragment j into buffer NXT, 8..15 = A fragment i.
    auto read_one = [&](auto buf, auto idx, uint32_t pa, uint32_t pb) {     // pa / pb: LDS byte addresses (32-bit: no base add at the use)
        constexpr int BUF = decltype(buf)::value, R = decltype(idx)::value;
        typedef const __attribute__((address_space(3))) frag *lfrag;
        if constexpr (BNN && (R < 8 || R >= 16)) {
            typedef short s16x4 __attribute__((ext_vector_type(4)));
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            constexpr int JB = R & 7;
            constexpr bool HI = R >= 16;                                                                               // k 4-7: block row a + 1, 2 KiB on
            const uint32_t q8 = (JB & 1) ? (pb ^ 32u) : pb;                                                            // odd column block: the other half
            const auto q = (__attribute__((address_space(3))) s16x4 *)(uintptr_t)(q8 + (uint32_t)((JB >> 1) * 256));
            const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(q + (HI ? 256 : 0));
            const s16x8 w = __builtin_shufflevector(v, v, 0, 1, 2, 3, 0, 1, 2, 3), cur = __builtin_bit_cast(s16x8, fb[BUF][JB]);
            if constexpr (HI) fb[BUF][JB] = __builtin_bit_cast(frag, (s16x8)__builtin_shufflevector(cur, w, 0, 1, 2, 3, 8, 9, 10, 11));
            else fb[BUF][JB] = __builtin_bit_cast(frag, (s16x8)__builtin_shufflevector(w, cur, 0, 1, 2, 3, 12, 13, 14, 15));
        } else if constexpr (R < 8) fb[BUF][R] = *(lfrag)(uintptr_t)(pb + (uint32_t)(R * 16 * ROW_BYTES));
        else fa[R - 8] = *(lfrag)(uintptr_t)(pa + (uint32_t)((R - 8) * 16 * ROW_BYTES));
    };
    auto dma_one = [&](auto is_b, auto jj, int64_t koff, char *base) {
        constexpr int J = decltype(jj)::value;
        if constexpr (decltype(is_b)::value && BNN) {    // koff = K-tile * 128 bytes along K = K-tile * 64 k-rows of ldb elements here
            glds16_s<J * 1024>(iss.ub + koff * g.ldb + (uint64_t)((uint32_t)(4 * (J >> 1)) * ldb_b) + (J & 1) * 256, (J & 4) ? voff_b1 : voff_b0, lds_addr_of(base));
        } else if constexpr (decltype(is_b)::value) {
            constexpr uint32_t ROWS = 32 * (J >> 2) + 16 * (J & 1) + 4 * ((J >> 1) & 1);
            glds16_s<J * 1024>(iss.ub + koff + (uint64_t)(ROWS * ldb_b), (J & 1) ? voff_b1 : voff_b0, lds_addr_of(base));
        } else {
            glds16_s<J * 1024>(iss.ua + koff + (uint64_t)((uint32_t)(8 * J) * lda_b), (J & 1) ? voff_a1 : voff_a0, lds_addr_of(base));
        }
    };
    // srcA = the B fragment (changes every MFMA), srcB = the A fragment (stays for eight): the order the matrix pipe issues at 16 cycles
    auto mfma_one = [&](auto buf, auto idx, auto first) {
        constexpr int BUF = decltype(buf)::value, I = decltype(idx)::value >> 3, J = decltype(idx)::value & 7;
        if constexpr (decltype(first)::value) qm<DT>::mfma0(fb[BUF][J], fa[I], acc[I][J]);
        else qm<DT>::mfma(fb[BUF][J], fa[I], acc[I][J]);
    };
    // (no instruction: keeps the 256 accumulators in the AGPR half next to 96 held registers, gemm_lp256q.hip)
    auto pin_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("" : "+a"(acc[i][j]));
    };
    // block IDX = (I = IDX >> 3, J = IDX & 7) is final: pack it.  Block rows 0-5 into P, 6-7 into the boundary staging image
    // (this wave's 8 KiB of the dead B slot: 32 rows x 256 B, 16-byte chunk ^ (row & 15)).
    int stage_off = 0;
    const uint32_t lx = BNN ? (uint32_t)((g4 >> 1) ^ l15) << 4 : (uint32_t)(g4 ^ l15) << 4;       // [N][K]: ((4 jj + g) ^ l15) << 4 = (jj << 6) ^ lx;
                                                                                                  // BNN: ((4 jj + 2 e + g / 2) ^ l15) << 4 = ((jj << 6) | (e << 5)) ^ lx, + 8 (g % 2)
    auto drain_one = [&](auto idx) {
        constexpr int I = decltype(idx)::value >> 3, J = decltype(idx)::value & 7, JJ = J >> 1, E = J & 1;
        uint32_t w0 = qm<DT>::pack2(acc[I][J][0], acc[I][J][1]), w1 = qm<DT>::pack2(acc[I][J][2], acc[I][J][3]);
        asm volatile("" : "+v"(w0), "+v"(w1));      // pins the packing HERE, between two MFMAs (left alone, hipcc sinks all 96 conversions of the held
                                                    // block rows to the tile boundary -- they have no side effect until the stores of the next tile)
        if constexpr (I >= 6) {
            const u32x2 w = {w0, w1};
            if constexpr (BNN) *reinterpret_cast<u32x2 *>(smem + stage_off + (I - 6) * 16 * 256 + ((((uint32_t)JJ << 6) | ((uint32_t)E << 5)) ^ opaque(lx))) = w;   // (stage_off carries 8 (g % 2))
            else *reinterpret_cast<u32x2 *>(smem + stage_off + (I - 6) * 16 * 256 + (((uint32_t)JJ << 6) ^ opaque(lx)) + 8 * E) = w;
        } else {
            P[I >> 1][I & 1][JJ][2 * E + 0] = w0;
            P[I >> 1][I & 1][JJ][2 * E + 1] = w1;
            if constexpr (BNN && E == 1) {      // both blocks of the pair are packed: rows 1, 3 of block 2 jj <-> rows 0, 2 of block 2 jj + 1 (see the kernel header)
                u32x4 &c4 = P[I >> 1][I & 1][JJ];
                const auto r0 = __builtin_amdgcn_permlane16_swap(c4[0], c4[2], false, false);
                const auto r1 = __builtin_amdgcn_permlane16_swap(c4[1], c4[3], false, false);
                c4[0] = r0[0]; c4[2] = r0[1]; c4[1] = r1[0]; c4[3] = r1[1];
            }
        }
    };

    // ---- where the HELD tile goes ---------------------------------------------------------------------------------------------
    // Chunk register jj of lane (l15, g) = row l15, bytes 64 jj + 16 g of the wave's 256-byte row.  Pair (ii, p) = chunk registers
    // 2 p, 2 p + 1 of block row ii: after the exchange with lane l15 ^ 1 register 2 p holds the EVEN row of the lane pair, register
    // 2 p + 1 the odd one, bytes 128 p + 64 (l15 & 1) + 16 g: one store instruction = 8 rows x one whole 128-byte line.
    char *hbase = nullptr;
    [[maybe_unused]] char *abl_base = nullptr;      // (QM_ABL & 4: the workgroup's first tile)
    bool held = false;           // P holds a finished tile whose stores are still to be issued (false only during a workgroup's first tile)
    const uint32_t pvoff = BNN ? (uint32_t)((l15 & ~1) * g.ldc * CSZ + 64 * (l15 & 1) + 32 * (g4 & 1) + 16 * (g4 >> 1))
                               : (uint32_t)((l15 & ~1) * g.ldc * CSZ + 64 * (l15 & 1) + 16 * g4);
    const int64_t rowbytes = g.ldc * CSZ, rowblock = (int64_t)32 * g.ldc * CSZ;
    const bool odd1 = (l15 & 1) != 0;
    // word W of pair TP (= 2 ii + p) of row block RB: 4 VALU
    auto transpose_word = [&](auto rbb, auto tpp, auto ww) {
        constexpr int RB = decltype(rbb)::value, II = decltype(tpp)::value >> 1, PP = decltype(tpp)::value & 1, W = decltype(ww)::value;
        const uint32_t a = P[RB][II][2 * PP][W], b = P[RB][II][2 * PP + 1][W];
        const uint32_t ax = (uint32_t)__builtin_amdgcn_mov_dpp((int)a, 0xB1, 0xF, 0xF, true);     // quad_perm [1,0,3,2]: lane ^ 1
        const uint32_t bx = (uint32_t)__builtin_amdgcn_mov_dpp((int)b, 0xB1, 0xF, 0xF, true);
        uint32_t lo = odd1 ? bx : a, hi = odd1 ? b : ax;
        asm volatile("" : "+v"(lo), "+v"(hi));          // (left alone, hipcc sinks the selects to the store that reads them: ten instructions in one MFMA gap)
        P[RB][II][2 * PP][W] = lo;
        P[RB][II][2 * PP + 1][W] = hi;
    };
    // store N (0..7) of row block RB: block row ii = N >> 2, line p = (N >> 1) & 1, odd rows = N & 1; rb_base = wave-uniform address of
    // the row block's first row
    auto store_one = [&](auto rbb, auto nn, char *rb_base) {
        constexpr int RB = decltype(rbb)::value, N = decltype(nn)::value, II = N >> 2, PP = (N >> 1) & 1, ODD = N & 1;
        const u32x4 v = P[RB][II][2 * PP + ODD];
        if (QM_ABL & 1) { asm volatile("" ::"v"(v)); return; }
        char *sb_ = rb_base + (II * 16 + ODD) * rowbytes;
        if (QM_ABL & 4) sb_ = abl_base + (II * 16 + ODD) * rowbytes;
#if QM_STORE_FORM == 1
        const uint32_t so_ = pvoff;          // (a local copy: a const captured only by an asm operand is not odr-used)
        // (the data registers are read after the issue and are dead behind it as far as hipcc knows -- it hands them to the next VALU result or
        //  LDS read at once: s_nop 1.  Without it the row-major-rhs form, whose next instruction is a v_xor, stored garbage: round 6)
        asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3" QM_NT_SUFFIX "\n\ts_nop 1" ::"v"(so_), "v"(v), "s"(sb_), "n"(PP * 128) : "memory");
#else
        __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(sb_ + PP * 128 + pvoff));
#endif
    };
    // the slots behind MFMAs 32 .. 63 of k-step 0 of the K-tile that carries store group G of the row block in P[0]: one word of a pair whose
    // first store is in the group takes two slots -- the two lane exchanges, then the two selects (four VALU instructions behind one MFMA
    // cost the K loop, two do not: profiles/r06_qm_pad_cost.txt)
    uint32_t tr_ax = 0, tr_bx = 0;
    auto tr_gap = [&](auto gg, auto nn) {
        constexpr int G = decltype(gg)::value, N = decltype(nn)::value;
        if constexpr (G >= 0 && N >= 32) {
            constexpr int S = N - 32, K = S >> 3, W = (S >> 1) & 3, H = S & 1;      // K-th pair of this K-tile, word W, half H
            constexpr int NPAIR = D >= 2 ? D / 2 : ((G & 1) == 0 ? 1 : 0);
            constexpr int FIRST_PAIR = D >= 2 ? G * (D / 2) : G / 2;
            if constexpr (K < NPAIR) {
                constexpr int TP = FIRST_PAIR + (K < NPAIR ? K : 0), II = TP >> 1, PP = TP & 1;
                const uint32_t a = P[0][II][2 * PP][W], b = P[0][II][2 * PP + 1][W];
                if constexpr (H == 0) {
                    tr_ax = (uint32_t)__builtin_amdgcn_mov_dpp((int)a, 0xB1, 0xF, 0xF, true);     // quad_perm [1,0,3,2]: lane ^ 1
                    tr_bx = (uint32_t)__builtin_amdgcn_mov_dpp((int)b, 0xB1, 0xF, 0xF, true);
                    asm volatile("" : "+v"(tr_ax), "+v"(tr_bx));
                } else {
                    uint32_t lo = odd1 ? tr_bx : a, hi = odd1 ? b : tr_ax;
                    asm volatile("" : "+v"(lo), "+v"(hi));
                    P[0][II][2 * PP][W] = lo;
                    P[0][II][2 * PP + 1][W] = hi;
                }
            }
        }
    };
    // the slots behind MFMAs 20, 22, 24, 28, 30, 32, 36, 38 of k-step 1 -- BEHIND the hand-over, between the B pieces: store i < D of group G.
    // (Until round 6 the stores sat in front of the hand-over and its wait allowed them to fly, vmcnt(8 + D) -- a count that differs between
    //  a workgroup's first tile and the others: two branches per K-tile.  Behind the hand-over the wait of the NEXT K-tile, vmcnt(8) = its own
    //  eight A pieces, covers them whether they exist or not.)
    auto st_gap = [&](auto gg, auto nn, char *rb_base) {
        constexpr int G = decltype(gg)::value, N = decltype(nn)::value;
        constexpr int I = N == 20 ? 0 : N == 22 ? 1 : N == 24 ? 2 : N == 28 ? 3 : N == 30 ? 4 : N == 32 ? 5 : N == 36 ? 6 : N == 38 ? 7 : -1;
        if constexpr (G >= 0 && I >= 0 && I < D) {
            if (held) store_one(IC<0>{}, IC<(G >= 0 ? G : 0) * D + (I >= 0 ? I : 0)>{}, rb_base);    // (uniform branch: the workgroup's first tile has nothing to store yet)
        }
    };

    // ---- block rows 6, 7 (the staging image) -------------------------------------------------------------------------------------
    // QM_BDRIP: the image is read back and stored 4 rows x 256 B at a time behind MFMA 8 it + 2 of k-step 0 of the NEXT tile's K-tile 0,
    // one MFMA in front of DMA piece `it` of that k-step -- the piece of unit 4 that overwrites exactly those 1 024 bytes of this wave's
    // region.  The read of part it + 1 is issued with the store of part it (two registers quads in rotation), so the store's operand wait
    // is what orders every read in front of the DMA piece that replaces its source.  vmcnt: the stores sit BETWEEN the pieces of unit 4
    // instead of in front of them: the same 16 operations may fly at hand-over 0.
    uint32_t bstage = 0;                 // LDS byte offset of this wave's staging image
    char *bbase = nullptr;               // wave-uniform address of row 96 of the wave's block
    const uint32_t bvoff = (uint32_t)((lane >> 4) * g.ldc * CSZ + (lane & 15) * 16);
    u32x4 bv[2];
    // row r = 4 it + lane / 16 of the image, chunk (lane % 16) ^ (r & 15)
    auto bread = [&](auto itt) {
        constexpr int IT = decltype(itt)::value;
        const uint32_t rdx = (uint32_t)((lane >> 4) * 256 + (((lane & 15) ^ (lane >> 4)) << 4));
        bv[IT & 1] = *reinterpret_cast<const u32x4 *>(smem + bstage + IT * 1024 + (opaque(rdx) ^ (uint32_t)((IT & 3) << 6)));
    };
    auto bstore = [&](auto itt) {
        constexpr int IT = decltype(itt)::value;
        const u32x4 v = bv[IT & 1];
        if (QM_ABL & 2) { asm volatile("" ::"v"(v)); return; }
        char *sb_ = bbase + (int64_t)(4 * IT) * rowbytes;
        const uint32_t so_ = bvoff;
        asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" ::"v"(so_), "v"(v), "s"(sb_) : "memory");
    };
    auto bd_gap = [&](auto nn) {
        constexpr int N = decltype(nn)::value;
        if constexpr (QM_BDRIP && (N & 7) == 2) {
            constexpr int IT = N >> 3;
            if (held) {
                bstore(IC<IT>{});
                if constexpr (IT < 7) bread(IC<(IT < 7 ? IT + 1 : 0)>{});
            }
        }
    };

    // 16 MFMAs n = N0 .. N0 + 15 of a k-step.  A read follows MFMA n where qm_read_at(k-step, n) >= 0 (below the kernel); DMASK bit b: a
    // DMA piece follows MFMA N0 + b.  KS = 0: transposition slots; KS = 1: store slots (head) + packing of finished blocks (DRAIN).
    // Instruction order pinned by sched_barrier after every group.
#define QM_G(CUR, NXT, KS, N0, BIT, FIRST, DRAIN, G)                                                                  \
    mfma_one(IC<CUR>{}, IC<(N0) + (BIT)>{}, IC<FIRST>{});                                                             \
    if constexpr (qm_read_at(KS, (N0) + (BIT), BNN) >= 0) read_one(IC<NXT>{}, IC<(qm_read_at(KS, (N0) + (BIT), BNN) >= 0 ? qm_read_at(KS, (N0) + (BIT), BNN) : 0)>{}, rd_a, rd_b); \
    gap_work(IC<KS>{}, IC<(N0) + (BIT)>{});                                                                           \
    if constexpr ((DRAIN) && (N0) + (BIT) >= QM_LAG) {                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
        drain_one(IC<((N0) + (BIT) >= QM_LAG ? (N0) + (BIT) - QM_LAG : 0)>{});                                        \
    }                                                                                                                 \
    if constexpr ((KS) == 0 && (G) >= 0) { __builtin_amdgcn_sched_barrier(0); tr_gap(IC<(G)>{}, IC<(N0) + (BIT)>{}); } \
    if constexpr ((KS) == 0 && (FIRST)) { __builtin_amdgcn_sched_barrier(0); bd_gap(IC<(N0) + (BIT)>{}); first_gap(IC<(N0) + (BIT)>{}); } \
    if constexpr ((KS) == 1 && (G) >= 0 && (N0) >= 16 && (N0) < 48) { __builtin_amdgcn_sched_barrier(0); st_gap(IC<(G)>{}, IC<(N0) + (BIT)>{}, rb_base); } \
    __builtin_amdgcn_sched_barrier(0);
#define QM_Q(CUR, NXT, KS, N0, FIRST, DRAIN, G)                                                                       \
    QM_G(CUR, NXT, KS, N0, 0, FIRST, DRAIN, G) QM_G(CUR, NXT, KS, N0, 1, FIRST, DRAIN, G) QM_G(CUR, NXT, KS, N0, 2, FIRST, DRAIN, G) QM_G(CUR, NXT, KS, N0, 3, FIRST, DRAIN, G) \
    QM_G(CUR, NXT, KS, N0, 4, FIRST, DRAIN, G) QM_G(CUR, NXT, KS, N0, 5, FIRST, DRAIN, G) QM_G(CUR, NXT, KS, N0, 6, FIRST, DRAIN, G) QM_G(CUR, NXT, KS, N0, 7, FIRST, DRAIN, G) \
    QM_G(CUR, NXT, KS, N0, 8, FIRST, DRAIN, G) QM_G(CUR, NXT, KS, N0, 9, FIRST, DRAIN, G) QM_G(CUR, NXT, KS, N0, 10, FIRST, DRAIN, G) QM_G(CUR, NXT, KS, N0, 11, FIRST, DRAIN, G) \
    QM_G(CUR, NXT, KS, N0, 12, FIRST, DRAIN, G) QM_G(CUR, NXT, KS, N0, 13, FIRST, DRAIN, G) QM_G(CUR, NXT, KS, N0, 14, FIRST, DRAIN, G) QM_G(CUR, NXT, KS, N0, 15, FIRST, DRAIN, G)

    // ---- first tile of this workgroup: units 0..3 (its K-tiles 0 and 1), then the first fragments --------------------------------
    uint32_t L = blockIdx.x;
    tile_src cur = locate(L);
    iss = cur;
    {
        const int64_t k0 = 0, k1 = ROW_BYTES;
        char *b0 = smem + dst_piece;
#define QM_PRO(IS_B, KOFF, SLOT)                                                                      \
        dma_one(IC<IS_B>{}, IC<0>{}, KOFF, b0 + SLOT * UNIT_BYTES); dma_one(IC<IS_B>{}, IC<1>{}, KOFF, b0 + SLOT * UNIT_BYTES); \
        dma_one(IC<IS_B>{}, IC<2>{}, KOFF, b0 + SLOT * UNIT_BYTES); dma_one(IC<IS_B>{}, IC<3>{}, KOFF, b0 + SLOT * UNIT_BYTES); \
        dma_one(IC<IS_B>{}, IC<4>{}, KOFF, b0 + SLOT * UNIT_BYTES); dma_one(IC<IS_B>{}, IC<5>{}, KOFF, b0 + SLOT * UNIT_BYTES); \
        dma_one(IC<IS_B>{}, IC<6>{}, KOFF, b0 + SLOT * UNIT_BYTES); dma_one(IC<IS_B>{}, IC<7>{}, KOFF, b0 + SLOT * UNIT_BYTES);
        QM_PRO(0, k0, 0) QM_PRO(1, k0, 1) QM_PRO(0, k1, 2) QM_PRO(1, k1, 3)
#undef QM_PRO
    }
    WAIT_VMCNT(16);                      // units 0, 1 landed (this wave's share)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    {
        const uint32_t rd_a = lds_addr_of(smem) + (uint32_t)ro_a, rd_b = lds_addr_of(smem) + (uint32_t)(UNIT_BYTES + ro_b);
        read_one(IC<0>{}, IC<0>{}, rd_a, rd_b); read_one(IC<0>{}, IC<8>{}, rd_a, rd_b); read_one(IC<0>{}, IC<1>{}, rd_a, rd_b); read_one(IC<0>{}, IC<2>{}, rd_a, rd_b);
        read_one(IC<0>{}, IC<3>{}, rd_a, rd_b); read_one(IC<0>{}, IC<4>{}, rd_a, rd_b); read_one(IC<0>{}, IC<5>{}, rd_a, rd_b); read_one(IC<0>{}, IC<6>{}, rd_a, rd_b);
        read_one(IC<0>{}, IC<7>{}, rd_a, rd_b); read_one(IC<0>{}, IC<9>{}, rd_a, rd_b); read_one(IC<0>{}, IC<10>{}, rd_a, rd_b); read_one(IC<0>{}, IC<11>{}, rd_a, rd_b);
        read_one(IC<0>{}, IC<12>{}, rd_a, rd_b); read_one(IC<0>{}, IC<13>{}, rd_a, rd_b); read_one(IC<0>{}, IC<14>{}, rd_a, rd_b); read_one(IC<0>{}, IC<15>{}, rd_a, rd_b);
        if constexpr (BNN) {
            read_one(IC<0>{}, IC<16>{}, rd_a, rd_b); read_one(IC<0>{}, IC<17>{}, rd_a, rd_b); read_one(IC<0>{}, IC<18>{}, rd_a, rd_b); read_one(IC<0>{}, IC<19>{}, rd_a, rd_b);
            read_one(IC<0>{}, IC<20>{}, rd_a, rd_b); read_one(IC<0>{}, IC<21>{}, rd_a, rd_b); read_one(IC<0>{}, IC<22>{}, rd_a, rd_b); read_one(IC<0>{}, IC<23>{}, rd_a, rd_b);
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    [[maybe_unused]] uint32_t qm_pad = 0, qm_padv = 0;
    int sa = 0;                          // ring byte offset of the A unit of the K-tile being multiplied
    int sb = UNIT_BYTES;                 // ... and of its B unit; the ring runs on across output tiles
    auto adv = [](int x, int n) { x += n * UNIT_BYTES; return x >= LDS_BYTES ? x - LDS_BYTES : x; };

    char *__restrict__ C = static_cast<char *>(g.c);
    uint32_t Lnext = 0;
    bool has_next = false;
    tile_src nxt = cur;
    int kbase = 0;                       // K-tile index of the issue side = t + 2 - kbase
    int t = 0;

    // ---- the K-tile's bookkeeping, one phase ahead, in MFMA gaps that carry nothing else --------------------------------------------------
    // Measured on config C3 (profiles/r06_qm_pad_cost.txt): a scalar or vector instruction behind an MFMA whose gap is empty, or holds one
    // other instruction, costs nothing; sixteen at the head of the K-tile (no MFMA in flight) cost 4 % = 6.4 cycles each, and a fifth
    // instruction in a gap that already held four as much.  Until this round a K-tile opened with 31 scalar instructions (ring arithmetic,
    // the issue side's tile switch and K offset, read and DMA addresses) and every DMA piece was five (64-bit base add, M0, the M0 hazard
    // nop, the load).  Now:
    //   * everything a K-tile's first half needs (pc_a, pc_m4, pc_ra0, pc_rb0; the tile switch and the K offset behind them) is computed in
    //     the previous K-tile's gaps behind MFMAs 0 .. 14 of k-step 1, what its second half needs (pc_b, pc_m5, pc_ra1, pc_rb1) in its own
    //     k-step 0: the registers are dead there (in-place, no copies), and a K-tile opens with its first MFMA;
    //   * a DMA piece is one v_add (its lane offset: the operand's + the piece's row offset, a gap earlier) and the load: the wave-uniform
    //     base of the unit stays in one SGPR pair, and M0 is written twice per unit instead of eight times -- the instruction offset of
    //     global_load_lds moves BOTH addresses (tools/dev/lds_dma_offset_probe.hip), so pieces J and J + 1 .. J + 3 share an M0 with offsets
    //     0 .. 3072 and the lane offset takes the 1 KiB steps back out (DMA_BIAS keeps it non-negative for every piece and layout).
#ifndef QM_PINMASK
#define QM_PINMASK 0xffffffffu
#endif
#define QM_PINS(i, x) do { if constexpr ((QM_PINMASK >> (i)) & 1u) asm volatile("" : "+s"(x)); } while (0)
    constexpr uint32_t DMA_BIAS = 3072;
    const char *pc_a = nullptr, *pc_b = nullptr;      // wave-uniform source of this K-tile's A pieces (unit 2t+4) / B pieces (unit 2t+5), less DMA_BIAS
    uint32_t pc_m4 = 0, pc_m5 = 0;                    // LDS byte address of this wave's first piece of those units
    uint32_t pc_ra0 = 0, pc_rb0 = 0;                  // ring offsets (per lane) of the fragment reads issued during k-step 0 (this K-tile's k-step 1) ...
    uint32_t pc_ra1 = 0, pc_rb1 = 0;                  // ... and behind the hand-over (K-tile t + 1's k-step 0)
    uint32_t koff = 0;                                // byte offset along K of the K-tile the issue side is at (K * 2 bytes < 2^32)
    uint32_t dma_vt = 0;                              // the lane offset of the next piece
    auto set_m0 = [&](uint32_t v) { asm volatile("s_mov_b32 m0, %0" ::"s"(v) : "memory"); };
    auto piece_off = [&](auto is_b, auto jj) -> uint32_t {      // (loop-invariant: 8 + 8 SGPRs, as before)
        constexpr int J = decltype(jj)::value;
        constexpr uint32_t ADJ = DMA_BIAS - (uint32_t)(J & 3) * 1024u;
        if constexpr (decltype(is_b)::value && BNN) return (uint32_t)(4 * (J >> 1)) * ldb_b + (uint32_t)((J & 1) * 256) + ADJ;
        else if constexpr (decltype(is_b)::value) return (uint32_t)(32 * (J >> 2) + 16 * (J & 1) + 4 * ((J >> 1) & 1)) * ldb_b + ADJ;
        else return (uint32_t)(8 * J) * lda_b + ADJ;
    };
    auto dma_prep = [&](auto is_b, auto jj) {
        constexpr int J = decltype(jj)::value;
        uint32_t v;
        if constexpr (decltype(is_b)::value && BNN) v = ((J & 4) ? voff_b1 : voff_b0) + piece_off(is_b, jj);
        else if constexpr (decltype(is_b)::value) v = ((J & 1) ? voff_b1 : voff_b0) + piece_off(is_b, jj);
        else v = ((J & 1) ? voff_a1 : voff_a0) + piece_off(is_b, jj);
        asm volatile("" : "+v"(v));
        dma_vt = v;
    };
    auto dma_go = [&](auto jj, const char *base) {
        constexpr int J = decltype(jj)::value;
        const uint32_t vt_ = dma_vt;         // (a local copy: a variable captured only by an asm operand is not odr-used)
        asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" ::"v"(vt_), "s"(base), "n"((J & 3) * 1024) : "memory");
    };
    // second-half state of THIS K-tile (k-step 0 gaps) and first-half state of the NEXT one (k-step 1 gaps in front of the hand-over)
    const uint32_t lds0 = lds_addr_of(smem);
    const uint32_t ra_k0 = lds0 + (uint32_t)ro_a, ra_k1 = lds0 + (uint32_t)(ro_a ^ 64);
    const uint32_t rb_k0 = lds0 + (uint32_t)ro_b, rb_k1 = lds0 + (uint32_t)(BNN ? ro_b + KS1_B : (ro_b ^ 64));
    const char *iss_a = iss.ua - DMA_BIAS, *iss_b = iss.ub - DMA_BIAS;     // the issue side's panels (iss), biased
    int pre_sw = 0;
    auto gap_work = [&](auto kss, auto nn) {
        constexpr int KS = decltype(kss)::value, N = decltype(nn)::value;
        if constexpr (KS == 0) {
            // (pure scalar work floats up to one gap ahead of its pin: odd N -- the gaps behind even MFMAs carry the fragment reads)
            if constexpr (N == 0) set_m0(pc_m4);
            if constexpr (N == 32) set_m0(pc_m4 + 4096);
            if constexpr ((N & 7) == 2) dma_prep(IC<0>{}, IC<(N >> 3)>{});
            if constexpr ((N & 7) == 3) dma_go(IC<(N >> 3)>{}, pc_a);
            if constexpr (N == 5) { pc_m5 = lds0 + (uint32_t)(sa + dst_piece); QM_PINS(4, pc_m5); }         // unit 2t+5 takes the slot of unit 2t
            if constexpr (N == 9) { sa = adv(sa, 2); QM_PINS(1, sa); }                                       // from here on: unit 2t+2
            if constexpr (N == 17) { pc_ra1 = (uint32_t)sa + ra_k0; asm volatile("" : "+v"(pc_ra1)); }
            if constexpr (N == 25) {
                if constexpr (BNN) pc_b = iss_b + (int64_t)koff * g.ldb; else pc_b = iss_b + koff;
                QM_PINS(3, pc_b);
            }
            if constexpr (N == 37) { pc_m4 = lds0 + (uint32_t)(sb + dst_piece); QM_PINS(11, pc_m4); }        // K-tile t + 1: unit 2t+6 takes the slot of unit 2t+1
            if constexpr (N == 41) { sb = adv(sb, 2); QM_PINS(2, sb); }                                      // from here on: unit 2t+3
            if constexpr (N == 45) { pc_rb1 = (uint32_t)sb + rb_k0; asm volatile("" : "+v"(pc_rb1)); }
        } else {
            // B unit 2t+5 behind the hand-over: pieces 0..3 behind MFMAs 19, 27, 35, 43, pieces 4..7 behind 49, 53, 57, 61
            if constexpr (N == 13) set_m0(pc_m5);
            if constexpr (N == 46) set_m0(pc_m5 + 4096);
            if constexpr (N >= 18 && N < 48 && (N & 7) == 2) dma_prep(IC<1>{}, IC<((N - 18) >> 3)>{});
            if constexpr (N >= 19 && N < 48 && (N & 7) == 3) dma_go(IC<((N - 19) >> 3)>{}, pc_b);
            if constexpr (N >= 48 && (N & 3) == 0) dma_prep(IC<1>{}, IC<(4 + ((N - 48) >> 2))>{});
            if constexpr (N >= 49 && (N & 3) == 1) dma_go(IC<(4 + ((N - 49) >> 2))>{}, pc_b);
            // K-tile t + 1
            if constexpr (N == 1) { pre_sw = (t + 1 == nk - 2 && has_next) ? 1 : 0; }
            if constexpr (N == 3) { iss_a = pre_sw ? nxt.ua - DMA_BIAS : iss_a; QM_PINS(6, iss_a); }
            if constexpr (N == 5) { iss_b = pre_sw ? nxt.ub - DMA_BIAS : iss_b; QM_PINS(7, iss_b); }
            if constexpr (N == 7) { kbase = pre_sw ? nk : kbase; QM_PINS(8, kbase); }
            if constexpr (N == 9) { koff = (uint32_t)min(t + 3 - kbase, nk - 1) * (uint32_t)ROW_BYTES; QM_PINS(9, koff); }   // clamp: only without a next tile
            if constexpr (N == 11) { pc_a = iss_a + koff; QM_PINS(10, pc_a); }
            if constexpr (N == 12) { pc_ra0 = (uint32_t)sa + ra_k1; asm volatile("" : "+v"(pc_ra0)); }
            if constexpr (N == 14) { pc_rb0 = (uint32_t)sb + rb_k1; asm volatile("" : "+v"(pc_rb0)); }
            if constexpr (N == 62) { ++t; QM_PINS(14, t); }
        }
    };
    // the state K-tile 0 of this workgroup's first tile starts from (iss = cur, kbase = 0)
    koff = 2u * ROW_BYTES;
    pc_a = iss_a + koff;
    pc_m4 = lds0 + (uint32_t)(adv(sa, 4) + dst_piece);
    pc_ra0 = (uint32_t)sa + ra_k1;
    pc_rb0 = (uint32_t)sb + rb_k1;

    // the tile after this one (branch-free: without one, the tile itself again -- its K-tiles 0 and 1 are fetched once more and never read)
    auto first_gap = [&](auto nn) {
        if constexpr (decltype(nn)::value == 40) {
            Lnext = L + gridDim.x;
            has_next = Lnext < total;
            nxt = locate(has_next ? Lnext : L);
        }
    };
    // one basic block per K-tile (gemm_lp256q.hip: hipcc schedules per block for register pressure and MFMAs carry no ordering edge)
#define QM_BLOCK_END() if (__builtin_expect(t > 0x3fffffff, 0)) asm volatile("s_trap 2");
    // One K-tile.  FIRST: K-tile 0 of an output tile (zero C operand in k-step 0).  LASTK: the tile's last K-tile (blocks are packed
    // as their last MFMA retires).  WAITN: LDS-DMA pieces + stores that may still fly at the hand-over when no group is stored here
    // (8: unit 2t+4; 16: + the 8 boundary stores in front of it).  G >= 0: store group G of the row block in P[0] -- its stores sit
    // in front of the hand-over and may fly too (vmcnt(8 + D)); everything older, the previous group included, has landed.
#define QM_KTILE(FIRST, LASTK, WAITN, G, BE)                                                                        \
    {                                                                                                               \
        uint32_t rd_a, rd_b;                                                                                        \
        rd_a = pc_ra0; rd_b = pc_rb0;                                                                 \
        QM_Q(0, 1, 0, 0, FIRST, 0, G) QM_Q(0, 1, 0, 16, FIRST, 0, G)                                                 \
        QM_Q(0, 1, 0, 32, FIRST, 0, G) QM_Q(0, 1, 0, 48, FIRST, 0, G)                                                \
        QM_Q(1, 0, 1, 0, 0, LASTK, G)                                                                                \
        QM_TW0()                                                                                                    \
        if constexpr ((WAITN) == 16) { if (held) WAIT_VMCNT(16); else WAIT_VMCNT(8); }                           \
        else WAIT_VMCNT(WAITN);                                                                                     \
        QM_TW1()                                                                                                    \
        WAIT_LGKM0();                    /* my reads of this K-tile are complete */                                  \
        __builtin_amdgcn_s_barrier();    /* BAR_t */                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
        QM_TW2()                                                                                                    \
        rd_a = pc_ra1; rd_b = pc_rb1;                                                                 \
        if constexpr (LASTK) stage_off = adv(sb, 3) + wave * 8192 + (int)opaque((uint32_t)(l15 * 256 + (BNN ? 8 * (g4 & 1) : 0)));   /* B unit of this K-tile (sb is unit 2t+3's by now): dead since BAR_t */ \
        QM_Q(1, 0, 1, 16, 0, LASTK, G) QM_Q(1, 0, 1, 32, 0, LASTK, G)                                                \
        QM_Q(1, 0, 1, 48, 0, LASTK, G)                                                                               \
        if constexpr (LASTK) {                                                                                      \
            drain_one(IC<64 - (QM_LAG >= 3 ? 3 : QM_LAG)>{}); drain_one(IC<64 - (QM_LAG >= 2 ? 2 : QM_LAG)>{}); drain_one(IC<63>{}); \
            __builtin_amdgcn_sched_barrier(0);                                                                      \
        }                                                                                                           \
        if constexpr (FIRST) pin_acc();                                                                             \
        if constexpr (BE) { QM_BLOCK_END() }                                                                        \
    }
    static_assert(QM_LAG >= 1 && QM_LAG <= 3, "the tail of the packing above covers up to three blocks");

#ifdef QM_TRACE
    int qt_tile = 0;
    unsigned long long qt_w[5] = {0, 0, 0, 0, 0};
#endif
    constexpr int GPR = 8 / D;           // K-tiles (store groups) per held row block
    for (;;) {
        // (the next tile's coordinates -- two divisions, ~100 scalar instructions -- are worked out behind MFMA 40 of this tile's first K-tile:
        //  first_gap, below; they are needed from K-tile nk - 3 on)
        kbase = 0;
        t = 0;
        char *rb_base = hbase;
        QM_STAMP(0);
        // ONE code path for every tile (the first tile of a workgroup runs the drip phase with its stores branched over: two paths --
        // held / not held -- met in front of the last K-tile with different register assignments, and hipcc bridged them with five
        // fragment spills + reloads behind an s_waitcnt vmcnt(0) at the top of every tile).
        // K-tile 0: the 8 boundary stores of the previous tile sit between unit 3 and unit 4 of this stream
        QM_KTILE(1, 0, 16, -1, 1)
        // drip phase, K-tiles 1 .. 24 / D: the row block in P[0] leaves, D stores per K-tile, then the next one moves down
#pragma nounroll
        for (int rb = 0; rb < 3; ++rb) {
            QM_KTILE(0, 0, 8, 0, 1)
            if constexpr (GPR > 1) QM_KTILE(0, 0, 8, 1, 1)
            if constexpr (GPR > 2) { QM_KTILE(0, 0, 8, 2, 1) QM_KTILE(0, 0, 8, 3, 1) }
            if constexpr (GPR > 4) { QM_KTILE(0, 0, 8, 4, 1) QM_KTILE(0, 0, 8, 5, 1) QM_KTILE(0, 0, 8, 6, 1) QM_KTILE(0, 0, 8, 7, 1) }
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) { P[0][ii][jj] = P[1][ii][jj]; P[1][ii][jj] = P[2][ii][jj]; }
            rb_base += rowblock;
        }
#pragma nounroll
        while (t < nk - 1) QM_KTILE(0, 0, 8, -1, 0)     // (the first of these waits for the last dripped stores: they are older than unit 2t+4)
        QM_KTILE(0, 1, 8, -1, 1)                                     // t == nk - 1: block rows 0..5 are packed into P, 6..7 into the staging image
        QM_STAMP(1);

        // ---- tile boundary: block rows 6, 7 through this wave's 8 KiB of the dead B slot ------------------------------------------
        {
            const int64_t cbase = cur.batch * g.stride_c;
            char *wbase = C + (cbase + (cur.m0 + wm * 128) * g.ldc + cur.n0 + wn * 128) * CSZ;   // my 128x128 block
            bstage = (uint32_t)(adv(sb, 3) + wave * 8192);          // slot of the last B unit, my DMA region of it
            bbase = wbase + (int64_t)96 * rowbytes;
            hbase = wbase;
            if ((QM_ABL & 4) && !held) abl_base = wbase;
            held = true;
            bread(IC<0>{});                                         // (same-wave hand-over: the DS ops of one wave execute in order)
            if constexpr (!QM_BDRIP) {
                bstore(IC<0>{}); bread(IC<1>{}); bstore(IC<1>{}); bread(IC<2>{}); bstore(IC<2>{}); bread(IC<3>{}); bstore(IC<3>{}); bread(IC<4>{});
                bstore(IC<4>{}); bread(IC<5>{}); bstore(IC<5>{}); bread(IC<6>{}); bstore(IC<6>{}); bread(IC<7>{}); bstore(IC<7>{});
                WAIT_LGKM0();                                       // staged rows are in registers before this wave's next DMA lands there
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        QM_STAMP(2);
#ifdef QM_TRACE
        ++qt_tile;
#endif
        if (!has_next) break;
        cur = nxt;
        L = Lnext;
    }
#undef QM_KTILE
    if (QM_PAD && (qm_pad == 0xdeadbeefu || qm_padv == 0xdeadbeefu)) asm volatile("s_trap 2");
#undef QM_Q
#undef QM_G
#ifdef QM_TRACE
    if (lane == 0 && blockIdx.x < 256)
        for (int k = 0; k < 5; ++k) qm_trace_buf[blockIdx.x * 64 + 32 + 4 * k + wave] = qt_w[k];
#endif
    // ---- the last tile of this workgroup has no K loop to hide under: its held stores leave at once ------------------------------
#define QM_FLUSH(RB)                                                                                                 \
    transpose_word(IC<RB>{}, IC<0>{}, IC<0>{}); transpose_word(IC<RB>{}, IC<0>{}, IC<1>{}); transpose_word(IC<RB>{}, IC<0>{}, IC<2>{}); transpose_word(IC<RB>{}, IC<0>{}, IC<3>{}); \
    transpose_word(IC<RB>{}, IC<1>{}, IC<0>{}); transpose_word(IC<RB>{}, IC<1>{}, IC<1>{}); transpose_word(IC<RB>{}, IC<1>{}, IC<2>{}); transpose_word(IC<RB>{}, IC<1>{}, IC<3>{}); \
    transpose_word(IC<RB>{}, IC<2>{}, IC<0>{}); transpose_word(IC<RB>{}, IC<2>{}, IC<1>{}); transpose_word(IC<RB>{}, IC<2>{}, IC<2>{}); transpose_word(IC<RB>{}, IC<2>{}, IC<3>{}); \
    transpose_word(IC<RB>{}, IC<3>{}, IC<0>{}); transpose_word(IC<RB>{}, IC<3>{}, IC<1>{}); transpose_word(IC<RB>{}, IC<3>{}, IC<2>{}); transpose_word(IC<RB>{}, IC<3>{}, IC<3>{}); \
    store_one(IC<RB>{}, IC<0>{}, hbase + (RB) * rowblock); store_one(IC<RB>{}, IC<1>{}, hbase + (RB) * rowblock);   \
    store_one(IC<RB>{}, IC<2>{}, hbase + (RB) * rowblock); store_one(IC<RB>{}, IC<3>{}, hbase + (RB) * rowblock);   \
    store_one(IC<RB>{}, IC<4>{}, hbase + (RB) * rowblock); store_one(IC<RB>{}, IC<5>{}, hbase + (RB) * rowblock);   \
    store_one(IC<RB>{}, IC<6>{}, hbase + (RB) * rowblock); store_one(IC<RB>{}, IC<7>{}, hbase + (RB) * rowblock);
    if constexpr (QM_BDRIP) {            // ... and neither have its block rows 6, 7 (part 0 is already in bv[0])
        bstore(IC<0>{}); bread(IC<1>{}); bstore(IC<1>{}); bread(IC<2>{}); bstore(IC<2>{}); bread(IC<3>{}); bstore(IC<3>{}); bread(IC<4>{});
        bstore(IC<4>{}); bread(IC<5>{}); bstore(IC<5>{}); bread(IC<6>{}); bstore(IC<6>{}); bread(IC<7>{}); bstore(IC<7>{});
    }
    QM_FLUSH(0) QM_FLUSH(1) QM_FLUSH(2)
#undef QM_FLUSH
    WAIT_VMCNT(0);                       // drain the clamped tail DMA (and the last stores) before the workgroup retires
}

template <int DT, int D, bool BNN>
void launch(mi355_ctx *ctx, hipStream_t s, const gemm_args &g, uint32_t batch)
{
    lds_opt_in(ctx, reinterpret_cast<const void *>(gemm_lp256qm_kernel<DT, D, BNN>), LDS_BYTES);
    const uint32_t total = g.tiles_m * g.tiles_n * batch;
    const uint32_t grid = std::min<uint32_t>(total, ctx->props.num_streaming_multiprocessors);   // one workgroup per CU (LDS admits no more)
    hipLaunchKernelGGL((gemm_lp256qm_kernel<DT, D, BNN>), dim3(grid), dim3(256), LDS_BYTES, s, g);
}

int drip_for(int64_t nk)                 // fewest stores per K-tile whose drip phase (K-tiles 1 .. 24 / D) fits: nk >= 24 / D + 3
{
    return nk >= 27 ? 1 : nk >= 15 ? 2 : nk >= 9 ? 4 : nk >= 6 ? 8 : 0;
}

}  // namespace

#ifdef QM_TRACE
extern "C" __attribute__((visibility("default"))) int mi355_dev_qm_trace(unsigned long long *host_out)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(qm_trace_buf), sizeof(unsigned long long) * 256 * 64);
}
#endif

namespace mi355 {

bool gemm_lp256qm_supports(const mi355_gemm_desc &d, const void *a, const void *b, const void *c)
{
    if (d.dtype_ab != MI355_DTYPE_BF16 && d.dtype_ab != MI355_DTYPE_F16) return false;
    if (d.dtype_c != d.dtype_ab) return false;                     // (f32 C would need 192 held registers)
    if (d.trans_a) return false;                                   // A [M][K]; B [N][K] or row-major [K][N]
    if (!d.trans_b && (int64_t)64 * d.ldb * 2 >= (1ll << 32)) return false;
    if (!gemm_lp256p_supports(d, a, b, c)) return false;           // full tiles, K-contiguous 16-byte aligned operands, 32-bit DMA offsets
    if (drip_for(d.k / 64) == 0) return false;
    if ((int64_t)32 * d.ldc * 2 >= (1ll << 32)) return false;      // per-lane store offsets are 32-bit
    return true;
}

int32_t launch_gemm_lp256qm(mi355_ctx *ctx, hipStream_t s, const mi355_gemm_desc &d, const void *a, const void *b, void *c)
{
    if (!gemm_lp256qm_supports(d, a, b, c))
        return fail(ctx, MI355_E_UNSUPPORTED, "lp256qm GEMM: shape/layout not supported by this kernel");
    gemm_args g{};
    g.a = a; g.b = b; g.c = c;
    g.m = d.m; g.n = d.n; g.k = d.k;
    g.lda = d.lda; g.ldb = d.ldb; g.ldc = d.ldc;
    g.stride_a = d.stride_a; g.stride_b = d.stride_b; g.stride_c = d.stride_c;
    g.tiles_m = (uint32_t)(d.m / BM);
    g.tiles_n = (uint32_t)(d.n / BN);
    g.group_m = 8;
    g.batch_count = (uint32_t)d.batch;
    const uint32_t batch = (uint32_t)d.batch;
    const int drip = drip_for(d.k / 64);
    const bool bf = d.dtype_ab == MI355_DTYPE_BF16;
#define QM_LAUNCH(DD) { if (d.trans_b) { if (bf) launch<MI355_DTYPE_BF16, DD, false>(ctx, s, g, batch); else launch<MI355_DTYPE_F16, DD, false>(ctx, s, g, batch); } \
                        else { if (bf) launch<MI355_DTYPE_BF16, DD, true>(ctx, s, g, batch); else launch<MI355_DTYPE_F16, DD, true>(ctx, s, g, batch); } }
    if (drip == 1) QM_LAUNCH(1) else if (drip == 2) QM_LAUNCH(2) else if (drip == 4) QM_LAUNCH(4) else QM_LAUNCH(8)
#undef QM_LAUNCH
    check_launch(ctx, "mi355_gemm(lp256qm)");
    return MI355_OK;
}

}  // namespace mi355
