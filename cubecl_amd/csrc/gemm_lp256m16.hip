// gemm_lp256m16.hip -- bf16 / f16 GEMM, 256 x 256 workgroup tile x one 128-byte K line (64 k-values), four waves, one per SIMD, on
// v_mfma_f32_16x16x32 (round 5).
//
// Roofline: MFMA bf16 / f16, 2.5 PFLOP/s dense.  Same tile, ring, LDS image, DMA addressing, hand-over and rasterisation as
// gemm_lp256w4.hip (read that header first); what differs is the matrix instruction and with it the fragment map and the epilogue.
//
// Why another shape.  On uniform[-1,1) operands the chip is power-limited: the 32x32x16 shape sustains 1 764 TFLOP/s at 1.80 GHz in a
// register-resident loop, the 16x16x32 shape -- half the accumulator bytes through the register file per FLOP -- holds 2.11-2.23 GHz.
// Round 2 measured the narrow shape 3 % BEHIND, because it issued at 18.7-19.4 cycles where 16 are nominal, and closed the question;
// tools/dev/mfma_issue_probe.hip (round 5) found the missing cycles: the instruction's SECOND source operand has to stay put across
// consecutive MFMAs -- eight MFMAs sharing srcB, srcA changing: 16.25 cycles; sharing srcA: 18.69; both changing: 20.6 -- and with
// that order the narrow shape reaches 1 946 TFLOP/s on uniform operands: +10 % over the wide one (profiles/r05_mfma_issue_probe.txt).
// This kernel is the 256 x 256 K loop built on it.
//
// Per wave: a 128 x 128 output = 8 x 8 blocks of 16 x 16 (256 accumulator registers, the AGPR half of the file).  A K-tile is two
// k-steps of 32; a k-step is 64 MFMAs, block row i (the A fragment, srcB) outer, block column j (the B fragment, srcA) inner.
// Fragments: lane (l15 = lane % 16, g = lane / 16) reads row l15 of a 16-row block, 16-byte chunk 4 s + g of the K-tile row -- one
// ds_read_b128, conflict free with the ring's swizzle (16 lanes of one g: rows 2r, 2r + 1 share chunk c ^ r and differ in bit 7 of the
// address).  16 reads per k-step, double-buffered in registers (128 VGPRs).
//
// Schedule of K-tile t (MFMA index n = 8 i + j within a k-step):
//     k-step 0: every other MFMA of 0-31 followed by one read of frags(t, 1);  two DMA pieces of unit 2t+4 (A of t+2) per 16 MFMAs
//     k-step 1: MFMA 0-15; vmcnt(8), lgkmcnt(0), s_barrier (BAR_t);  every other MFMA of 16-47 followed by one read of frags(t+1, 0),
//               the eight DMA pieces of unit 2t+5 (B of t+2, into the slot of unit 2t: dead since BAR_t) spread over MFMA 16-63
//   Same counts as the 32x32 kernel per K-tile and wave (128 MFMA x 16 cycles = 2 048 pipe cycles, 32 ds_read_b128, 16 DMA pieces,
//   one barrier), same vmcnt reasoning.  The last two K-tiles run a copy of the body without DMA.
//
// Restrictions: 16-bit operands, A row-major [M][K], B stored [N][K], K % 64 == 0, 16-byte aligned operand rows; C f32 or the operand
// type, any alignment (element-wise stores off the 16-byte grid); M, N arbitrary (edge tiles clamp their loads, skip their stores).
#include <algorithm>
#include <type_traits>

#include "gemm_common.hpp"

using namespace mi355;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef M16_GROUP_M
#define M16_GROUP_M 8   // tile rows per rasterisation group (as gemm_lp256w4.hip: each XCD's 32 resident tiles form an 8 x 4 patch); 4 / 16 measured: see profiles/r05_m16_schedule.txt
#endif
constexpr int BM = 256, BN = 256;
constexpr int ROW_BYTES = 128;
constexpr int UNIT_BYTES = BM * ROW_BYTES;        // 32 KiB: one ring slot
constexpr int NSLOT = 5;
constexpr int LDS_BYTES = NSLOT * UNIT_BYTES;     // 160 KiB

template <int DT> struct m16;
template <> struct m16<MI355_DTYPE_BF16> {
    typedef bf16x8 frag;
    static __device__ __forceinline__ f32x4 mfma(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct m16<MI355_DTYPE_F16> {
    typedef f16x8 frag;
    static __device__ __forceinline__ f32x4 mfma(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

// LDS-DMA: 64-bit wave-uniform base in SGPRs + 32-bit per-lane byte offset, LDS destination through M0 (gemm_lp256w4.hip glds16_s)
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
#define M16_STR_(x) #x
#define M16_STR(x) M16_STR_(x)
#define WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" M16_STR(n) ")" ::: "memory")
#define WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
template <int V> using IC = std::integral_constant<int, V>;

template <int DT, int DT_C>
__global__ void __launch_bounds__(256)
gemm_lp256m16_kernel(gemm_args g)
{
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    typedef typename m16<DT>::frag frag;
    constexpr int ESZ = 2, BK = 64;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int g4 = lane >> 4, l15 = lane & 15;

    uint32_t tm, tn, batch_u;
    batched_tile_coords(g.tiles_m, g.tiles_n, g.group_m, tm, tn, batch_u);      // XCD remap over the (batch, tile) sequence
    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
    const int64_t batch = batch_u;
    const char *__restrict__ A = static_cast<const char *>(g.a) + batch * g.stride_a * ESZ;
    const char *__restrict__ B = static_cast<const char *>(g.b) + batch * g.stride_b * ESZ;
    const int nk = (int)(g.k / BK);

    // ---- DMA map (as gemm_lp256w4.hip): a unit is 32 pieces of 1 KiB (8 rows); this wave fills pieces wave * 8 + j
    const int sub = lane >> 3, c8 = lane & 7;
    const char *ubase_a = A + m0 * g.lda * ESZ;
    const char *ubase_b = B + n0 * g.ldb * ESZ;
    uint32_t voff_a[8], voff_b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int r = wave * 64 + j * 8 + sub;
        const int q = c8 ^ ((r >> 1) & 7);
        voff_a[j] = (uint32_t)(min((int64_t)r, g.m - 1 - m0) * g.lda * ESZ + q * 16);
        voff_b[j] = (uint32_t)(min((int64_t)r, g.n - 1 - n0) * g.ldb * ESZ + q * 16);
    }
    const int dst_piece = wave * 8 * 1024;

    // ---- fragment read offsets: row * 128 + ((4 s + g) ^ f) * 16, f = (row >> 1) & 7 = (l15 >> 1) & 7 for every 16-row block
    const int f = (l15 >> 1) & 7;
    const int rowoff_a = (wm * 128 + l15) * ROW_BYTES, rowoff_b = (wn * 128 + l15) * ROW_BYTES;
    const int x0 = ((g4) ^ f) << 4, x1 = ((4 + g4) ^ f) << 4;

    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    frag fa[2][8], fb[2][8];

    // read R of a k-step, in order of first use by the next k-step's MFMAs (i outer, j inner): b0, a0, b1 .. b7, a1 .. a7
    auto read_one = [&](auto buf, auto idx, const char *pa, const char *pb) {
        constexpr int BUF = decltype(buf)::value, R = decltype(idx)::value;
        if constexpr (R == 0) fb[BUF][0] = *reinterpret_cast<const frag *>(pb);
        else if constexpr (R == 1) fa[BUF][0] = *reinterpret_cast<const frag *>(pa);
        else if constexpr (R <= 8) fb[BUF][R - 1] = *reinterpret_cast<const frag *>(pb + (R - 1) * 16 * ROW_BYTES);
        else fa[BUF][R - 8] = *reinterpret_cast<const frag *>(pa + (R - 8) * 16 * ROW_BYTES);
    };
    auto dma_one = [&](auto is_b, auto jj, int64_t koff, char *base) {
        constexpr int J = decltype(jj)::value;
        glds16_s<J * 1024>((decltype(is_b)::value ? ubase_b : ubase_a) + koff, decltype(is_b)::value ? voff_b[J] : voff_a[J], lds_addr_of(base));
    };
    // srcA = the B fragment (changes every MFMA), srcB = the A fragment (stays for eight): the order the matrix pipe issues at 16 cycles
    auto mfma_one = [&](auto buf, auto idx) {
        constexpr int BUF = decltype(buf)::value, I = decltype(idx)::value >> 3, J = decltype(idx)::value & 7;
        acc[I][J] = m16<DT>::mfma(fb[BUF][J], fa[BUF][I], acc[I][J]);
    };

    // 16 MFMAs n = N0 .. N0 + 15 of a k-step; RMASK / DMASK bit b: a read / a DMA piece follows MFMA N0 + b.  Reads are numbered
    // R0, R0 + 1, .. and pieces J0, J0 + 1, .. in mask order.  Instruction order pinned by sched_barrier after every group.
#define M16_G(CUR, NXT, N0, BIT, RMASK, R0, DMASK, IS_B, J0)                                                       \
    mfma_one(IC<CUR>{}, IC<(N0) + (BIT)>{});                                                                      \
    if constexpr (((RMASK) >> (BIT)) & 1u) read_one(IC<NXT>{}, IC<(R0) + __builtin_popcount((RMASK) & ((1u << (BIT)) - 1u))>{}, rd_a, rd_b); \
    if constexpr (((DMASK) >> (BIT)) & 1u) dma_one(IC<IS_B>{}, IC<(J0) + __builtin_popcount((DMASK) & ((1u << (BIT)) - 1u))>{}, dma_koff, dma_base); \
    __builtin_amdgcn_sched_barrier(0);
#define M16_Q(CUR, NXT, N0, RMASK, R0, DMASK, IS_B, J0)                                                            \
    M16_G(CUR, NXT, N0, 0, RMASK, R0, DMASK, IS_B, J0) M16_G(CUR, NXT, N0, 1, RMASK, R0, DMASK, IS_B, J0)           \
    M16_G(CUR, NXT, N0, 2, RMASK, R0, DMASK, IS_B, J0) M16_G(CUR, NXT, N0, 3, RMASK, R0, DMASK, IS_B, J0)           \
    M16_G(CUR, NXT, N0, 4, RMASK, R0, DMASK, IS_B, J0) M16_G(CUR, NXT, N0, 5, RMASK, R0, DMASK, IS_B, J0)           \
    M16_G(CUR, NXT, N0, 6, RMASK, R0, DMASK, IS_B, J0) M16_G(CUR, NXT, N0, 7, RMASK, R0, DMASK, IS_B, J0)           \
    M16_G(CUR, NXT, N0, 8, RMASK, R0, DMASK, IS_B, J0) M16_G(CUR, NXT, N0, 9, RMASK, R0, DMASK, IS_B, J0)           \
    M16_G(CUR, NXT, N0, 10, RMASK, R0, DMASK, IS_B, J0) M16_G(CUR, NXT, N0, 11, RMASK, R0, DMASK, IS_B, J0)         \
    M16_G(CUR, NXT, N0, 12, RMASK, R0, DMASK, IS_B, J0) M16_G(CUR, NXT, N0, 13, RMASK, R0, DMASK, IS_B, J0)         \
    M16_G(CUR, NXT, N0, 14, RMASK, R0, DMASK, IS_B, J0) M16_G(CUR, NXT, N0, 15, RMASK, R0, DMASK, IS_B, J0)

    // ---- prologue: units 0 .. 3 (K-tiles 0 and 1), then the fragments of (0, 0) ------------------------------------------------
    {
        const int64_t k0 = 0, k1 = (int64_t)min(1, nk - 1) * ROW_BYTES;   // nk == 1: K-tile 0 twice, drained at the hand-over
        char *b0 = smem + dst_piece;
#define M16_PRO(IS_B, KOFF, SLOT)                                                                                   \
        dma_one(IC<IS_B>{}, IC<0>{}, KOFF, b0 + SLOT * UNIT_BYTES); dma_one(IC<IS_B>{}, IC<1>{}, KOFF, b0 + SLOT * UNIT_BYTES); \
        dma_one(IC<IS_B>{}, IC<2>{}, KOFF, b0 + SLOT * UNIT_BYTES); dma_one(IC<IS_B>{}, IC<3>{}, KOFF, b0 + SLOT * UNIT_BYTES); \
        dma_one(IC<IS_B>{}, IC<4>{}, KOFF, b0 + SLOT * UNIT_BYTES); dma_one(IC<IS_B>{}, IC<5>{}, KOFF, b0 + SLOT * UNIT_BYTES); \
        dma_one(IC<IS_B>{}, IC<6>{}, KOFF, b0 + SLOT * UNIT_BYTES); dma_one(IC<IS_B>{}, IC<7>{}, KOFF, b0 + SLOT * UNIT_BYTES);
        M16_PRO(0, k0, 0) M16_PRO(1, k0, 1) M16_PRO(0, k1, 2) M16_PRO(1, k1, 3)
#undef M16_PRO
    }
    WAIT_VMCNT(16);                      // units 0, 1 landed (this wave's share)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    {
        const char *rd_a = smem + rowoff_a + x0, *rd_b = smem + UNIT_BYTES + rowoff_b + x0;
        read_one(IC<0>{}, IC<0>{}, rd_a, rd_b); read_one(IC<0>{}, IC<1>{}, rd_a, rd_b); read_one(IC<0>{}, IC<2>{}, rd_a, rd_b); read_one(IC<0>{}, IC<3>{}, rd_a, rd_b);
        read_one(IC<0>{}, IC<4>{}, rd_a, rd_b); read_one(IC<0>{}, IC<5>{}, rd_a, rd_b); read_one(IC<0>{}, IC<6>{}, rd_a, rd_b); read_one(IC<0>{}, IC<7>{}, rd_a, rd_b);
        read_one(IC<0>{}, IC<8>{}, rd_a, rd_b); read_one(IC<0>{}, IC<9>{}, rd_a, rd_b); read_one(IC<0>{}, IC<10>{}, rd_a, rd_b); read_one(IC<0>{}, IC<11>{}, rd_a, rd_b);
        read_one(IC<0>{}, IC<12>{}, rd_a, rd_b); read_one(IC<0>{}, IC<13>{}, rd_a, rd_b); read_one(IC<0>{}, IC<14>{}, rd_a, rd_b); read_one(IC<0>{}, IC<15>{}, rd_a, rd_b);
    }
    __builtin_amdgcn_sched_barrier(0);

    int sa = 0;                          // ring byte offset of unit 2t   (A of K-tile t)
    int sb = UNIT_BYTES;                 // ring byte offset of unit 2t+1 (B of K-tile t)
    auto adv = [](int x, int n) { x += n * UNIT_BYTES; return x >= LDS_BYTES ? x - LDS_BYTES : x; };

    // Where a K-tile's 32 fragment reads, 16 DMA pieces and its hand-over sit among its 128 MFMAs (dev switch M16_SCHED; measured on
    // 8192^3, profiles/r05_m16_schedule.txt):
    //   0  k-step 0: a read behind every other MFMA of 0-31, two pieces of unit 2t+4 per 16 MFMAs; k-step 1: hand-over behind MFMA 15,
    //      a read behind every other MFMA of 16-47, unit 2t+5 spread over 16-63
    //   1  as 0 with the reads of k-step 0 behind every FOURTH MFMA of 0-63
    //   2  hand-over behind MFMA 31 of k-step 1, its reads behind every other MFMA of 32-63
    //   3  (the first version) all 16 reads of a k-step behind 16 consecutive MFMAs, hand-over behind MFMA 31
#ifndef M16_SCHED
#define M16_SCHED 0
#endif
#if M16_SCHED == 1
#define M16_STEP0(ISSUE) M16_Q(0, 1, 0, 0x1111u, 0, (ISSUE) ? 0x0404u : 0u, 0, 0) M16_Q(0, 1, 16, 0x1111u, 4, (ISSUE) ? 0x0404u : 0u, 0, 2) \
                         M16_Q(0, 1, 32, 0x1111u, 8, (ISSUE) ? 0x0404u : 0u, 0, 4) M16_Q(0, 1, 48, 0x1111u, 12, (ISSUE) ? 0x0404u : 0u, 0, 6)
#elif M16_SCHED == 3
#define M16_STEP0(ISSUE) M16_Q(0, 1, 0, 0xFFFFu, 0, 0u, 0, 0) M16_Q(0, 1, 16, 0u, 0, (ISSUE) ? 0xAAAAu : 0u, 0, 0) M16_Q(0, 1, 32, 0u, 0, 0u, 0, 0) M16_Q(0, 1, 48, 0u, 0, 0u, 0, 0)
#else
#define M16_STEP0(ISSUE) M16_Q(0, 1, 0, 0x5555u, 0, (ISSUE) ? 0x0808u : 0u, 0, 0) M16_Q(0, 1, 16, 0x5555u, 8, (ISSUE) ? 0x0808u : 0u, 0, 2) \
                         M16_Q(0, 1, 32, 0u, 0, (ISSUE) ? 0x0808u : 0u, 0, 4) M16_Q(0, 1, 48, 0u, 0, (ISSUE) ? 0x0808u : 0u, 0, 6)
#endif
#if M16_SCHED == 2
#define M16_STEP1_HEAD M16_Q(1, 0, 0, 0u, 0, 0u, 0, 0) M16_Q(1, 0, 16, 0u, 0, 0u, 0, 0)
#define M16_STEP1_TAIL(ISSUE) M16_Q(1, 0, 32, 0x5555u, 0, (ISSUE) ? 0x2222u : 0u, 1, 0) M16_Q(1, 0, 48, 0x5555u, 8, (ISSUE) ? 0x2222u : 0u, 1, 4)
#elif M16_SCHED == 3
#define M16_STEP1_HEAD M16_Q(1, 0, 0, 0u, 0, 0u, 0, 0) M16_Q(1, 0, 16, 0u, 0, 0u, 0, 0)
#define M16_STEP1_TAIL(ISSUE) M16_Q(1, 0, 32, 0xFFFFu, 0, 0u, 0, 0) M16_Q(1, 0, 48, 0u, 0, (ISSUE) ? 0x5555u : 0u, 1, 0)
#else
#define M16_STEP1_HEAD M16_Q(1, 0, 0, 0u, 0, 0u, 0, 0)
#define M16_STEP1_TAIL(ISSUE) M16_Q(1, 0, 16, 0x5555u, 0, (ISSUE) ? 0x0808u : 0u, 1, 0) M16_Q(1, 0, 32, 0x5555u, 8, (ISSUE) ? 0x0808u : 0u, 1, 2) \
                              M16_Q(1, 0, 48, 0u, 0, (ISSUE) ? 0x2222u : 0u, 1, 4)
#endif
    // One K-tile.  ISSUE = 1: the steady state; ISSUE = 0: the last two K-tiles (nothing left to fetch, the hand-over waits for all).
#define M16_KTILE(ISSUE)                                                                                           \
    {                                                                                                             \
        const int sa1 = adv(sa, 2), sb1 = adv(sb, 2);     /* units 2t+2, 2t+3 (K-tile t+1) */                     \
        const int s4 = adv(sa, 4);                        /* unit 2t+4 -> slot of unit 2t-1 */                     \
        const int s5 = sa;                                /* unit 2t+5 -> slot of unit 2t   */                     \
        const int64_t dma_koff = (int64_t)(t + 2) * ROW_BYTES;                                                    \
        const char *rd_a, *rd_b;                                                                                  \
        char *dma_base;                                                                                           \
        /* k-step 0 (buffer 0): the 16 reads of k-step 1 behind every other MFMA of 0-31, unit 2t+4 two pieces per 16 MFMAs.   */ \
        /* (Reads and pieces are spread thin: a 16-cycle MFMA hides one LDS or DMA issue, not a run of them -- with the 16      */ \
        /* reads behind 16 consecutive MFMAs the loop ran at 0.76 of the MFMA rate at its clock, profiles/r05_m16_schedule.txt) */ \
        rd_a = smem + sa + rowoff_a + x1; rd_b = smem + sb + rowoff_b + x1; dma_base = smem + s4 + dst_piece;       \
        M16_STEP0(ISSUE)                                                                                          \
        M16_STEP1_HEAD                                                                                            \
        if (ISSUE) WAIT_VMCNT(8); else WAIT_VMCNT(0);     /* my share of K-tile t+1 landed; unit 2t+4 may fly */    \
        WAIT_LGKM0();                                     /* my reads of K-tile t are complete */                  \
        __builtin_amdgcn_s_barrier();                     /* BAR_t */                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        rd_a = smem + sa1 + rowoff_a + x0; rd_b = smem + sb1 + rowoff_b + x0; dma_base = smem + s5 + dst_piece;     \
        M16_STEP1_TAIL(ISSUE)                                                                                     \
        sa = sa1;                                                                                                 \
        sb = sb1;                                                                                                 \
    }
    int t = 0;
    for (; t + 2 < nk; ++t) M16_KTILE(1)
    for (; t < nk; ++t) M16_KTILE(0)
#undef M16_KTILE
#undef M16_Q
#undef M16_G
    // nothing is in flight here: the last hand-over waited for vmcnt(0) and no DMA was issued after it

    // ---- epilogue ------------------------------------------------------------------------------------------------------------
    // With srcA = the B fragment, lane (l15, g) holds for block (i, j): C[m = 16 i + l15][n = 16 j + 4 g .. + 3] -- four consecutive
    // columns of ONE row.  Each wave transposes its 128 x 128 block through its own LDS scratch, 32 rows (two block rows) at a time,
    // and writes whole rows: 16 bytes per lane, 256 (16-bit) / 512 (f32) contiguous bytes per row (as gemm_lp256w4.hip).
    char *__restrict__ C = static_cast<char *>(g.c);
    constexpr int CSZ = (DT_C == MI355_DTYPE_F32) ? 4 : 2;
    const int64_t cbase = batch * g.stride_c;
    {
        constexpr int RS = 128 * CSZ + 16;                 // staged row pitch in bytes
        constexpr int STAGE = 32 * RS;
        constexpr int LPR = 128 * CSZ / 16;                // lanes per output row: 16 / 32
        constexpr int RPI = 64 / LPR;                      // rows per store instruction: 4 / 2
        __builtin_amdgcn_s_barrier();                      // every wave is done with the ring
        char *stage = smem + wave * ((STAGE + 1023) & ~1023);
        char *wr = stage + l15 * RS + 4 * g4 * CSZ;
        const char *rd = stage + (lane / LPR) * RS + (lane % LPR) * 16;
        char *crow = C + (cbase + (m0 + wm * 128 + lane / LPR) * g.ldc + n0 + wn * 128) * CSZ + (lane % LPR) * 16;
        const int64_t cstep = (int64_t)RPI * g.ldc * CSZ;
        constexpr int EPP = 16 / CSZ;
        const int64_t row0 = m0 + wm * 128 + lane / LPR;
        const int64_t col0 = n0 + wn * 128 + (lane % LPR) * EPP;
        const int ncols = (int)max((int64_t)0, min((int64_t)EPP, g.n - col0));
        const bool cvec = ((((uint64_t)g.ldc * CSZ) | ((uint64_t)g.stride_c * CSZ) | reinterpret_cast<uintptr_t>(g.c)) & 15u) == 0;
        const bool interior = cvec && (m0 + BM <= g.m) && (n0 + BN <= g.n);
#pragma unroll
        for (int ip = 0; ip < 4; ++ip) {                   // block rows 2 ip, 2 ip + 1 = 32 rows of the wave's block
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const f32x4 v = acc[2 * ip + ii][j];
                    char *d = wr + ii * 16 * RS + j * 16 * CSZ;
                    if constexpr (DT_C == MI355_DTYPE_F32) *reinterpret_cast<f32x4 *>(d) = v;
                    else if constexpr (DT_C == MI355_DTYPE_BF16) { bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]}; *reinterpret_cast<bf16x4 *>(d) = o; }
                    else { f16x4 o = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]}; *reinterpret_cast<f16x4 *>(d) = o; }
                }
            WAIT_LGKM0();                                  // same-wave hand-over: DS ops of one wave execute in order
            char *cdst = crow + (int64_t)ip * 32 * g.ldc * CSZ;
#pragma unroll
            for (int it = 0; it < 32 / RPI; ++it) {
                const u32x4 v = *reinterpret_cast<const u32x4 *>(rd + it * RPI * RS);
                if (!interior) {
                    if (row0 + ip * 32 + it * RPI >= g.m || ncols <= 0) continue;
                    if (ncols < EPP || !cvec) {
#pragma unroll
                        for (int e = 0; e < EPP; ++e) {
                            if (e >= ncols) break;
                            if constexpr (CSZ == 4) reinterpret_cast<float *>(cdst + it * cstep)[e] = __uint_as_float(v[e]);
                            else reinterpret_cast<uint16_t *>(cdst + it * cstep)[e] = (uint16_t)(v[e >> 1] >> ((e & 1) * 16));
                        }
                        continue;
                    }
                }
                __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(cdst + it * cstep));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int DT, int DT_C>
void launch(mi355_ctx *ctx, hipStream_t s, const gemm_args &g, uint32_t batch)
{
    lds_opt_in(ctx, reinterpret_cast<const void *>(gemm_lp256m16_kernel<DT, DT_C>), LDS_BYTES);
    hipLaunchKernelGGL((gemm_lp256m16_kernel<DT, DT_C>), dim3(g.tiles_m * g.tiles_n, batch), dim3(256), LDS_BYTES, s, g);
}

}  // namespace

namespace mi355 {

bool gemm_lp256m16_supports(const mi355_gemm_desc &d, const void *a, const void *b, const void *c)
{
    if (d.dtype_ab != MI355_DTYPE_BF16 && d.dtype_ab != MI355_DTYPE_F16) return false;
    if (d.trans_a || !d.trans_b) return false;
    return gemm_lp256w4_supports(d, a, b, c);          // same tile, same operand rules
}

int32_t launch_gemm_lp256m16(mi355_ctx *ctx, hipStream_t s, const mi355_gemm_desc &d, const void *a, const void *b, void *c)
{
    if (!gemm_lp256m16_supports(d, a, b, c))
        return fail(ctx, MI355_E_UNSUPPORTED, "lp256m16 GEMM: shape/layout not supported by this kernel");
    gemm_args g{};
    g.a = a; g.b = b; g.c = c;
    g.m = d.m; g.n = d.n; g.k = d.k;
    g.lda = d.lda; g.ldb = d.ldb; g.ldc = d.ldc;
    g.stride_a = d.stride_a; g.stride_b = d.stride_b; g.stride_c = d.stride_c;
    g.tiles_m = (uint32_t)((d.m + BM - 1) / BM);
    g.tiles_n = (uint32_t)((d.n + BN - 1) / BN);
    g.group_m = M16_GROUP_M;
    const uint32_t batch = (uint32_t)d.batch;
    constexpr int BF = MI355_DTYPE_BF16, HF = MI355_DTYPE_F16, CF = MI355_DTYPE_F32;
    if (d.dtype_ab == BF) {
        if (d.dtype_c == CF) launch<BF, CF>(ctx, s, g, batch); else launch<BF, BF>(ctx, s, g, batch);
    } else {
        if (d.dtype_c == CF) launch<HF, CF>(ctx, s, g, batch); else launch<HF, HF>(ctx, s, g, batch);
    }
    check_launch(ctx, "mi355_gemm(lp256m16)");
    return MI355_OK;
}

}  // namespace mi355
