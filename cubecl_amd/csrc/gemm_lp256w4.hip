// gemm_lp256w4.hip -- bf16 / f16 / f32 / fp8 GEMM, 256x256 workgroup tile x one 128-byte K line (128 fp8,
// 64 16-bit or 32 f32 k-values), FOUR waves, one per SIMD.
//
// Roofline: MFMA bf16/f16, ~2.5 PFLOP/s dense; MFMA f32 (v_mfma_f32_32x32x2_f32, exact f32), 157.3 TFLOP/s;
// MFMA fp8 (v_mfma_f32_32x32x64_f8f6f4, OCP e4m3 / e5m2), ~5 PFLOP/s dense (MI355X_MICROARCH.md).  This is the headline kernel for BASELINE configs C3 (8192^3 bf16), C5 (batched
// 2048^3 bf16) and C2 (4096^3 f32, K-contiguous operands).
//
// Why one wave per SIMD.  The 8-wave ping-pong kernel (gemm_lp256.hip of rounds 1-4, retired in round 5) handed the matrix pipe of a
// SIMD back and forth between two waves through s_barrier; its ablations (profiles/) show the
// hand-over itself costs ~16 % with every load removed, and the load phases barely overlap the
// MFMA phases.  Here each SIMD hosts ONE wave that owns a 128 x 128 output (4 x 4 MFMA tiles of
// 32x32x16, 256 accumulator registers in the AGPR half of the unified 512-entry file) and never
// gives the pipe away: fragment reads and LDS-DMA issues are slotted BETWEEN its own MFMAs (an
// MFMA occupies the pipe for 32 cycles; up to ~5 other instructions issue for free meanwhile --
// MI355X_MICROARCH.md "one wave per SIMD").  Per K-tile and wave: 64 MFMA, 32 ds_read_b128,
// 16 LDS-DMA, ONE s_barrier.  LDS fragment traffic drops from 192 KiB to 128 KiB per K-tile
// against the 2x4 wave grid (each operand half is read by 2 waves instead of 4 / 2).
//
// LDS: 160 KiB = ring of 5 slots x 32 KiB.  "Unit" u = 2t is the A tile (256 rows x 128 B) of
// K-tile t, u = 2t+1 its B tile; unit u lives in slot u % 5.  Rows are one 128-byte line (64
// k-values); filled by LDS-DMA (global_load_lds_dwordx4: 1 KiB = 8 rows per wave instruction) with
// the XOR swizzle on the SOURCE address and on the fragment read (conflict-free ds_read_b128, as
// gemm_lp128.hip).
//
// Schedule of K-tile t (k-steps s = 0..3 of 16 MFMAs; fragments double-buffered in registers):
//     s=0: read frags(t,1)   DMA unit 2t+4, pieces 0-3      MFMA(t,0)
//     s=1: read frags(t,2)   DMA unit 2t+4, pieces 4-7      MFMA(t,1)
//     s=2: read frags(t,3)                                   MFMA(t,2)
//          vmcnt(8): my share of units <= 2t+3 (K-tile t+1) has landed; lgkmcnt(0): my reads of
//          K-tile t are complete;   s_barrier  (BAR_t)
//     s=3: read frags(t+1,0) DMA unit 2t+5, pieces 0-7      MFMA(t,3)
//   After BAR_t every wave's K-tile t+1 data is visible and nobody reads K-tile t's slots again,
//   so unit 2t+5 may overwrite slot (2t+5)%5 == slot of unit 2t, and the first fragments of
//   K-tile t+1 are fetched under the last 16 MFMAs of K-tile t: the barrier is the only point
//   where the pipe can drain.  Unit 2t+4 reuses the slot of unit 2t-1 (dead since BAR_{t-1}).
//   Every DMA is issued >= 3 k-steps (~1500 cycles) before the barrier that needs it; vmcnt
//   never reaches 0 in the steady state.  The last two K-tiles of a tile have nothing left to fetch: they run a
//   second copy of the body without DMA and with vmcnt(0) at the hand-over.
//
// Restrictions (the dispatcher falls back to gemm_lp128.hip or the re-layout pass otherwise):
//   K % 64 (16-bit) / 32 (f32) == 0, A row-major [M][K], B stored [N][K] (trans_b = 1; f32 also takes row-major
//   [K][N] with N % 4 == 0), operand and C rows 16-byte aligned.  M and N are arbitrary: edge tiles clamp their
//   loads to the last valid row and skip the stores outside the matrix.
#include <algorithm>
#include <type_traits>

#include "gemm_common.hpp"

using namespace mi355;

namespace {

constexpr int BM = 256, BN = 256;
constexpr int ROW_BYTES = 128;                    // one K-tile row = one 128-byte line: 64 x 16-bit or 32 x f32
constexpr int UNIT_BYTES = BM * ROW_BYTES;        // 32 KiB: one ring slot
constexpr int NSLOT = 5;
constexpr int LDS_BYTES = NSLOT * UNIT_BYTES;     // 160 KiB

// A "fragment" is the 16 bytes one lane reads per 32-row block and k-step: 8 x 16-bit values feeding ONE
// v_mfma_f32_32x32x16, or 4 x f32 feeding FOUR v_mfma_f32_32x32x2_f32 (element c of the A and of the B
// fragment go to MFMA c: lane-half h then supplies k = 8s + 4h + c for both operands, so every k of the
// K-tile is used exactly once -- only the order of the exact-f32 accumulation changes).
template <int DT> struct lp;
template <> struct lp<MI355_DTYPE_BF16> {
    typedef bf16x8 frag;
    static constexpr int ESZ = 2;
    static __device__ __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c)
    { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct lp<MI355_DTYPE_F16> {
    typedef f16x8 frag;
    static constexpr int ESZ = 2;
    static __device__ __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c)
    { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
// fp8: one v_mfma_f32_32x32x64_f8f6f4 (64 cycles, twice the bf16 rate) eats 32 bytes per lane and operand.  Measured
// with structured scales (tools/dev/mx_diag2.py): registers 0-3 of BOTH lane-halves form the first 32 k-values of the
// 64-wide step (lanes 0-31: k 0..15, lanes 32-63: k 16..31; scaled, in the MX form, by the scale lanes 0-31 supply),
// registers 4-7 the second 32 (scale from lanes 32-63).  Unscaled, any consistent k permutation is as good as another,
// so that kernel simply feeds lane-half h two adjacent 16-byte chunks.  A K-tile (128 k-values) is two such steps.  The instruction is emitted without block scales (all scale operands zero selects
// the unscaled encoding; checked in the ISA); cbsz / blgp = 0 reads e4m3, 1 reads e5m2.
template <> struct lp<MI355_DTYPE_F8E4M3> {
    typedef i32x8 frag;
    static constexpr int ESZ = 1;
    static __device__ __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c)
    { return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0); }
};
template <> struct lp<MI355_DTYPE_F8E5M2> {
    typedef i32x8 frag;
    static constexpr int ESZ = 1;
    static __device__ __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c)
    { return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 1, 1, 0, 0, 0, 0); }
};
// MX fp4 (e2m1, two per byte, first element in the low nibble): a lane's 16 bytes are 32 k-values = one MX block =
// its whole share of a 32x32x64 step, which then takes 32 cycles (4x the bf16 rate).  The host hands the kernel the
// BYTE matrix (k, lda, ldb, strides halved), so ESZ = 1 and a K-tile (128 bytes) is 256 k-values = 4 k-steps: the
// bf16 schedule as is.  fp4 exists only block-scaled (there is no unscaled fp4 matrix instruction).
template <> struct lp<MI355_DTYPE_F4E2M1X2> {
    typedef i32x4 frag;
    static constexpr int ESZ = 1;
    static __device__ __forceinline__ f32x16 mfma(frag, frag, f32x16 c) { return c; }   // never used unscaled
};
// cbsz / blgp operand-format codes of v_mfma_scale_f32_32x32x64_f8f6f4
template <int DT> struct mx_fmt { static constexpr int value = DT == MI355_DTYPE_F8E4M3 ? 0 : DT == MI355_DTYPE_F8E5M2 ? 1 : 4; };
__device__ __forceinline__ i32x8 widen(i32x8 v) { return v; }
__device__ __forceinline__ i32x8 widen(i32x4 v) { i32x8 w = {v[0], v[1], v[2], v[3], 0, 0, 0, 0}; return w; }   // fp4: the upper half is not read
// D = C + (X .* 2^(sx-127)) (Y .* 2^(sy-127)): lane l scales its own 32 k-values with byte OPSEL of its scale register
template <int FMT_X, int FMT_Y, int OPSEL, typename FX, typename FY>
__device__ __forceinline__ f32x16 mfma_mx(FX x, FY y, f32x16 c, uint32_t sx, uint32_t sy)
{
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(widen(x), widen(y), c, FMT_X, FMT_Y, OPSEL, (int)sx, OPSEL, (int)sy);
}
template <> struct lp<MI355_DTYPE_F32> {
    typedef f32x4 frag;
    static constexpr int ESZ = 4;
    static __device__ __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c)
    {
#pragma unroll
        for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], c, 0, 0, 0);
        return c;
    }
};

#ifndef W4_DMA_AUX
#define W4_DMA_AUX 0   // cache-policy bits of the LDS-DMA loads (dev: 2 = nt)
#endif
#ifndef W4_GROUP_M
#define W4_GROUP_M 8   // tile rows per rasterisation group: each XCD's 32 resident tiles form a GROUP_M x 32/GROUP_M patch
#endif
__device__ __forceinline__ void glds16(const void *gsrc, void *lds_dst)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)lds_dst, 16, 0, W4_DMA_AUX);
}

// LDS-DMA with the address split the way the hardware wants it: 64-bit wave-uniform base in SGPRs + 32-bit
// per-lane byte offset, LDS destination (wave-uniform) through M0.  Written as inline asm because hipcc
// re-associates base + offset into per-lane 64-bit pointers (one 64-bit VALU add per piece in the K loop).
// M0 is set in the same statement that uses it (guide 5.7); nothing else in this kernel touches M0.  The
// compiler does not count these loads: every wait on them is an explicit counted s_waitcnt in this file.
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

#ifndef W4_ABL
#define W4_ABL 0          // dev ablations: 1 no DMA, 2 no fragment reads, 4 no MFMA, 8 no hand-over barrier, 16 no C stores (results: profiles/r01_power_ablation.md)
#endif

#define WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" W4_STR(n) ")" ::: "memory")
#ifndef W4_NT_C
#define W4_NT_C 1     // 1: non-temporal C stores (+1 % at 8192^3, neutral at 4096^3) (keep A/B rather than C in the 256 MiB Infinity Cache)
#endif
#define W4_STR_(x) #x
#define W4_STR(x) W4_STR_(x)
#define WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
template <int V> using IC = std::integral_constant<int, V>;

// Dev timing trace (-DW4_TRACE): per-workgroup shader-clock stamps {kernel entry, first MFMA, loop end, kernel end}
// of wave 0, read back with mi355_dev_w4_trace (dev builds only; never in the product library).
#ifdef W4_TRACE
__device__ unsigned long long w4_trace_buf[4096 * 8];
#define W4_STAMP(k) do { if (tid == 0 && blockIdx.x < 4096 && blockIdx.y == 0) { w4_trace_buf[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); \
        if ((k) == 0 || (k) == 3) w4_trace_buf[blockIdx.x * 8 + 4 + ((k) == 3)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define W4_STAMP(k)
#endif

// BNN: B is row-major [K][N] instead of [N][K] -- the layout TensorHandle::new_contiguous gives a rhs
// (crates/cubecl-std/src/tensor/handle.rs:89; the reference's row-major cmma test, runtime_tests/cmma.rs:1160-1177).
//   f32: the K-tile is 32 k-rows of 256 n-values (1 KiB each = one DMA piece, no swizzle), and a B fragment is four
//   ds_read_b32 (consecutive lanes -> consecutive n: conflict free) instead of one ds_read_b128 -- affordable because an
//   f32 k-step holds 64 MFMAs of 64 cycles.
//   bf16 / f16 (round 3; was: a transposition pass into library scratch): the K-tile is 64 k-rows of 256 n-values = 512 B
//   each.  The matrix core wants, per lane, 8 consecutive k of ONE column -- strided by the row pitch in this image -- which
//   is what ds_read_b64_tr_b16 delivers: the 16 lanes of a group hand in the addresses of a [4 k][16 n] block (lane i: row
//   i/4, columns 4(i%4)..+3) and lane l receives column l, k 0..3.  LDS-DMA writes lane-linear but READS from any per-lane
//   address, so the image is built for that gather at no cost: 128 blocks of [4 k][32 n] = 256 contiguous bytes (block
//   (a, b) = k-rows 4a..4a+3 x columns 32b..32b+31 at (a*8 + b)*256, row i of it at +64 i), i.e. a 32-lane read touches one
//   whole 256-byte bank row: conflict free by construction.  One DMA piece (1 KiB) = four blocks of one a = 4 k-rows x 256
//   contiguous bytes of global memory (two whole lines per row).  A B fragment is two tr reads (k 0..3 and 4..7 of the lane-
//   half's eight: blocks a = 4s + 2h and + 1, 2 KiB apart), issued in the slot the ds_read_b128 had.
// MX (block-scaled, one ue8m0 scale per 32 k-values; DTB = B's element type, fp8 formats may be mixed): the scales come
// pre-arranged by gemm_scaled.cpp as ST[K-tile][row padded to the tile grid][NB bytes] (NB = 4 fp8 / 8 fp4 blocks per
// K-tile row), so a lane's share is one coalesced 4-byte load per 32-row block and K-tile (fp4: the four blocks of its
// lane-half; fp8: the row's four blocks, its own two shifted into place), issued one K-tile ahead into a second register set.  These loads are inline asm like the
// DMA (the compiler must not insert its own waits into the counted vmcnt stream).
__device__ __forceinline__ void scale_ld32(uint32_t &dst, const void *ubase, uint32_t voff, int imm)
{
    asm volatile("global_load_dword %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(ubase), "n"(imm) : "memory");
}

// NJ (round 5): 32-column fragments per wave -- 4 = the 256 x 256 tile; 3 = a 256 x 192 tile (each wave 128 x 96, 12 MFMAs per
// k-step, 192 accumulators) for grids that leave CUs idle with the square tile (3072^3: 144 tiles of 256 x 256 on 256 CUs, 192 of
// 256 x 192).  Same ring, same schedule: a B unit is 24 KiB in its 32 KiB slot (6 DMA pieces per wave instead of 8), the reads,
// MFMAs and pieces of column fragment 3 are simply not emitted.  Plain [N][K] 16-bit operands only.
// NI: the same for the rows -- 3 = 192 tile rows (each wave 96 rows): with NJ = 3 a 192 x 192 tile, 9 MFMAs per k-step, 144 accumulators.
template <int DT, int DT_C, bool BNN = false, int DTB = DT, bool MX = false, bool ATN = false, int NJ = 4, int NI = 4>
__global__ void __launch_bounds__(256)
gemm_lp256w4_kernel(gemm_args g)
{
    static_assert((NJ == 4 && NI == 4) || ((NJ == 3 || NJ == 4) && (NI == 3 || NI == 4) && !MX && !ATN && (DT == MI355_DTYPE_BF16 || DT == MI355_DTYPE_F16)),
                  "narrow tiles: 16-bit operands, A [M][K], B [N][K] or row-major [K][N]");
    constexpr int BNT = NJ * 64;                        // tile columns: 256 or 192
    constexpr int NPB = NJ * 2;                         // DMA pieces of a B unit per wave: 8 or 6
    constexpr int BMT = NI * 64;                        // tile rows: 256 or 192
    constexpr int NPA = NI * 2;                         // DMA pieces of an A unit per wave
    static_assert(!BNN || DT == MI355_DTYPE_F32 || DT == MI355_DTYPE_BF16 || DT == MI355_DTYPE_F16, "row-major B: f32 and 16-bit operands");
    constexpr bool BNN16 = BNN && (DT == MI355_DTYPE_BF16 || DT == MI355_DTYPE_F16);
    // ATN (late round 3): A stored [K][M] together with a row-major B (lhs^T . grad_out) -- the A tile of a K-tile is 64 k-rows x 256 m,
    // the mirror image of the row-major B tile: same blocks, same pieces, same transposing reads (gemm_lp128.hip ATN)
    static_assert(!ATN || BNN16, "A stored [K][M]: 16-bit operands, together with a row-major B");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    typedef typename lp<DT>::frag frag;
    static_assert(sizeof(typename lp<DTB>::frag) == sizeof(frag), "A and B fragments must have the same width");

    const int tid = threadIdx.x;
    W4_STAMP(0);
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5, l31 = lane & 31;

    uint32_t tm, tn, batch_u;
    batched_tile_coords(g.tiles_m, g.tiles_n, g.group_m, tm, tn, batch_u);      // XCD remap over the (batch, tile) sequence
    const int64_t m0 = (int64_t)tm * BMT, n0 = (int64_t)tn * BNT;
    const int64_t batch = batch_u;
    constexpr int ESZ = lp<DT>::ESZ;
    constexpr bool F8 = DT == MI355_DTYPE_F8E4M3 || DT == MI355_DTYPE_F8E5M2;
    constexpr bool F4 = DT == MI355_DTYPE_F4E2M1X2;
    static_assert(!F4 || MX, "fp4 exists only block-scaled");
    static_assert(!MX || F8 || F4, "block scaling is an fp8 / fp4 feature");
    constexpr int BK = ROW_BYTES / ESZ;                 // 128 (fp8) / 64 (16-bit) / 32 (f32) k-values per K-tile
    const char *__restrict__ A = static_cast<const char *>(g.a) + batch * g.stride_a * ESZ;
    const char *__restrict__ B = static_cast<const char *>(g.b) + batch * g.stride_b * ESZ;
    const int nk = (int)(g.k / BK);

    // ---- DMA map: a unit is 32 pieces of 1 KiB (8 rows); this wave fills pieces wave*8 + j.
    //   lane -> (row = piece*8 + lane/8, physical chunk c = lane%8), source chunk = c ^ ((row>>1)&7).
    //   ((row>>1)&7 depends on j only through its parity, so two per-lane pointers per operand
    //   suffice; the (j>>1) step is a wave-uniform byte offset.)
    const int sub = lane >> 3, c8 = lane & 7;
    // Addresses are split into a wave-uniform 64-bit base (kernel arguments, tile, wave: SGPRs, advanced with
    // scalar adds) and a 32-bit per-lane byte offset that never changes, so that each DMA is
    // `global_load_lds_dwordx4 v_off, s[base:base+1]` with no 64-bit vector add per piece.
    // Ragged edges: rows past M (N) are clamped to the last valid row, so edge tiles read valid memory and compute
    // garbage in accumulator rows/columns the epilogue never stores.
    const char *ubase_a = A + m0 * g.lda * ESZ;                           // uniform: first row of the tile
    const char *ubase_b = B + n0 * g.ldb * ESZ;
    uint32_t voff_a[8], voff_b[8];                                        // piece j of this wave: rows j*8 + lane/8
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int r = wave * (BMT / 4) + j * 8 + sub;                     // tile row of this lane's 16 bytes (BMT rows: NPA pieces per wave)
        const int q = c8 ^ ((r >> 1) & 7);
        voff_a[j] = (uint32_t)(min((int64_t)r, g.m - 1 - m0) * g.lda * ESZ + q * 16);
        const int rb = wave * (BNT / 4) + j * 8 + sub;                    // (the B tile has BNT rows: NPB pieces per wave)
        const int qb = c8 ^ ((rb >> 1) & 7);
        voff_b[j] = (uint32_t)(min((int64_t)rb, g.n - 1 - n0) * g.ldb * ESZ + qb * 16);
    }
    // BNN, f32: piece j of this wave is k-row wave*8 + j of the K-tile, 256 n-values = 64 lanes x 16 B
    // BNN, 16-bit: piece p = wave*8 + j holds blocks (a = p/2, b = 4(p%2) .. +3): lane -> block b = 4(p%2) + lane/16, row
    //   i = (lane%16)/4 of it, 16-byte chunk lane%4 = columns 32b + 8(lane%4) .. +7 of k-row 4a + i
    const char *ubase_bnn = B + (int64_t)(BNN16 ? wave * 16 : wave * 8) * g.ldb * ESZ + n0 * ESZ;
    uint32_t voff_bnn[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {                                         // columns past N re-read the last valid 16 bytes
        if constexpr (BNN16) {
            // blocks of a block row: 2 NJ (8, or 6 on the 192-column tile: a piece's four blocks then straddle block rows)
            const int blk = 4 * j + (lane >> 4), a_local = blk / (2 * NJ), bcol = blk % (2 * NJ);
            const int64_t col = bcol * 32 + (lane & 3) * 8;
            voff_bnn[j] = (uint32_t)((a_local * 4 + ((lane & 15) >> 2)) * g.ldb * ESZ + min(col, g.n - 8 - n0) * ESZ);
        } else
            voff_bnn[j] = (uint32_t)(j * g.ldb * ESZ + min((int64_t)lane * 4, g.n - 4 - n0) * ESZ);
    }
    const char *ubase_atn = A + (int64_t)(wave * 16) * g.lda * ESZ + m0 * ESZ;
    uint32_t voff_atn[ATN ? 8 : 1];
    if constexpr (ATN) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {                                     // rows of C past M re-read the last valid 16 bytes
            const int64_t col = (j & 1) * 128 + (lane >> 4) * 32 + (lane & 3) * 8;
            voff_atn[j] = (uint32_t)(((j >> 1) * 4 + ((lane & 15) >> 2)) * g.lda * ESZ + min(col, g.m - 8 - m0) * ESZ);
        }
    }
    const int dst_piece = wave * 8 * 1024;                            // + j*1024 within the slot
    constexpr int DST_PIECE_B_STEP = (NPB - 8) * 1024;                // B units of the 192-column tile: wave * NPB pieces in
    constexpr int DST_PIECE_A_STEP = (NPA - 8) * 1024;                // ... and A units of the 192-row tile

    // ---- fragment read offsets: row*128 + ((2s+h) ^ f) * 16, f = (row>>1)&7 = (l31>>1)&7 for every tile row
    const int f = (l31 >> 1) & 7;
    // fp8: the second 16 bytes of a fragment sit in physical chunk ^ 1 (unscaled: logical chunks 2c, 2c+1); MX: logical
    // chunks c and c+2 (registers 0-3 of both lane-halves are one MX block, registers 4-7 the next) = physical ^ 2
    const int hd = MX ? ((f & 2) ? -32 : 32) : ((f & 1) ? -16 : 16);
    const int rowoff_a = ATN ? wm * 4 * 256 + ((lane & 15) >> 2) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8 : (wm * (NI * 32) + l31) * ROW_BYTES;
    // (16-bit row-major B: block b = 4 wn + j of the lane-half's block row, + row (lane%16)/4, + 16-lane group, + 8 B per lane)
    constexpr int BROW = 2 * NJ * 256;                                // bytes of one block row of the row-major B image (8 or 6 blocks)
    const int rowoff_b = BNN16 ? wn * NJ * 256 + ((lane & 15) >> 2) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8
                         : BNN ? (wn * 128 + l31) * 4 : (wn * (NJ * 32) + l31) * ROW_BYTES;

    f32x16 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    frag fa[2][4];
    typename lp<DTB>::frag fb[2][4];

    // ---- MX scales: sc_* = this K-tile's (byte OPSEL = k-step), sn_* = the next K-tile's, in flight --------------
    constexpr int NB = F4 ? 8 : 4;                                        // MX blocks (scale bytes) per K-tile row
    uint32_t sc_a[4] = {0, 0, 0, 0}, sc_b[4] = {0, 0, 0, 0}, sn_a[4] = {0, 0, 0, 0}, sn_b[4] = {0, 0, 0, 0};
    const char *sbase_a = nullptr, *sbase_b = nullptr;                    // uniform: ST[t] + tile row 0 (+ batch)
    uint32_t svoff_a = 0, svoff_b = 0;
    const int sshift = F4 ? 0 : 8 * h;
    int64_t sstep_a = 0, sstep_b = 0;                                     // bytes from ST[t] to ST[t+1]
    if constexpr (MX) {
        sstep_a = (int64_t)g.tiles_m * BM * NB;
        sstep_b = (int64_t)g.tiles_n * BN * NB;
        sbase_a = static_cast<const char *>(g.sa) + batch * g.stride_sa + m0 * NB;
        sbase_b = static_cast<const char *>(g.sb) + batch * g.stride_sb + n0 * NB;
        // fp4: a lane-half owns the 4 blocks (bytes) 4h .. 4h+3 of the K-tile row; fp8: every lane fetches the row's
        // 4 bytes and shifts its own into place when it takes them (lanes 0-31 scale blocks 0 / 2, lanes 32-63 blocks 1 / 3)
        svoff_a = (uint32_t)((wm * 128 + l31) * NB + (F4 ? 4 * h : 0));
        svoff_b = (uint32_t)((wn * 128 + l31) * NB + (F4 ? 4 * h : 0));
    }
    // scale load number Q of a K-tile: Q = 0..3 -> A row-block Q, 4..7 -> B row-block Q-4 (32 rows = 32*NB bytes apart)
    auto scale_one = [&](auto qq) {
        constexpr int Q = decltype(qq)::value;
        if constexpr (MX) {
            if constexpr (Q < 4) scale_ld32(sn_a[Q & 3], sbase_a, svoff_a, (Q & 3) * 32 * NB);
            else scale_ld32(sn_b[Q & 3], sbase_b, svoff_b, (Q & 3) * 32 * NB);
        }
    };
    // the scales of the next K-tile have landed (at most NEWER younger vector-memory operations may still fly): take them
#define W4_TAKE_SCALES(NEWER)                                                                                  \
    if constexpr (MX) {                                                                                        \
        asm volatile("s_waitcnt vmcnt(" W4_STR(NEWER) ")"                                                       \
                     : "+v"(sn_a[0]), "+v"(sn_a[1]), "+v"(sn_a[2]), "+v"(sn_a[3]), "+v"(sn_b[0]), "+v"(sn_b[1]),   \
                       "+v"(sn_b[2]), "+v"(sn_b[3])::"memory");                                                 \
        /* fp8: byte 0 / 2 of the shifted word = block h / 2+h of the K-tile = this lane-half's scale in k-step 0 / 1 */ \
        _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) { sc_a[q_] = sn_a[q_] >> sshift; sc_b[q_] = sn_b[q_] >> sshift; } \
        sbase_a += sstep_a;                                                                                    \
        sbase_b += sstep_b;                                                                                    \
    }

    // fragment load order == order of first use by the next k-step's MFMAs (j outer, i inner)
    auto read_one = [&](auto buf, auto idx, const char *pa, const char *pb) {
        constexpr int BUF = decltype(buf)::value, R = decltype(idx)::value;
        if (W4_ABL & 2) return;
        if constexpr (F8) {
            // fp8: 16 reads per k-step of 64; read R fills half R&1 of fragment R>>1 (same fragment order as below)
            constexpr int FR = R >> 1, HALF = R & 1;
            const char *p = (FR == 0) ? pb : (FR <= 4) ? pa + (FR - 1) * 32 * ROW_BYTES : pb + (FR - 4) * 32 * ROW_BYTES;
            const u32x4 v = *reinterpret_cast<const u32x4 *>(p + (HALF ? hd : 0));
            frag &dst = (FR == 0) ? fb[BUF][0] : (FR <= 4) ? fa[BUF][(FR - 1) & 3] : fb[BUF][(FR - 4) & 3];
            dst[4 * HALF + 0] = (int)v[0]; dst[4 * HALF + 1] = (int)v[1]; dst[4 * HALF + 2] = (int)v[2]; dst[4 * HALF + 3] = (int)v[3];
        } else if constexpr (BNN16) {
            if constexpr (R >= 1 && R <= 4) {
                if constexpr (R - 1 >= NI) { /* the 192-row tile: three row blocks */ }
                else if constexpr (ATN) {                             // row block R - 1: k 0..3 from block row a, k 4..7 from a + 1
                    typedef short s16x4 __attribute__((ext_vector_type(4)));
                    typedef short s16x8 __attribute__((ext_vector_type(8)));
                    const auto q = (const __attribute__((address_space(3))) s16x4 *)(pa + (R - 1) * 256);
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(const_cast<__attribute__((address_space(3))) s16x4 *>(q));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(const_cast<__attribute__((address_space(3))) s16x4 *>(q + 256));
                    const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    fa[BUF][R - 1] = __builtin_bit_cast(frag, both);
                } else
                    fa[BUF][R - 1] = *reinterpret_cast<const frag *>(pa + (R - 1) * 32 * ROW_BYTES);
            } else {
                constexpr int JB = (R == 0) ? 0 : R - 4;       // column block JB: k 0..3 from block row a, k 4..7 from a + 1
                if constexpr (JB < NJ) {
                typedef short s16x4 __attribute__((ext_vector_type(4)));
                typedef short s16x8 __attribute__((ext_vector_type(8)));
                const auto q = (const __attribute__((address_space(3))) s16x4 *)(pb + JB * 256);
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(const_cast<__attribute__((address_space(3))) s16x4 *>(q));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(const_cast<__attribute__((address_space(3))) s16x4 *>(q + BROW / 8));
                const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                fb[BUF][JB] = __builtin_bit_cast(typename lp<DTB>::frag, both);
                }
            }
        } else if constexpr (BNN) {
            if (R >= 1 && R <= 4) fa[BUF][R - 1] = *reinterpret_cast<const frag *>(pa + (R - 1) * 32 * ROW_BYTES);
            else {
                constexpr int JB = (R == 0) ? 0 : R - 4;       // element e of B fragment JB: k-row e of this lane-half's four
#pragma unroll
                for (int e = 0; e < 4; ++e) fb[BUF][JB][e] = *reinterpret_cast<const float *>(pb + e * 1024 + JB * 128);
            }
        } else {
            if (R == 0) fb[BUF][0] = *reinterpret_cast<const frag *>(pb);
            else if (R <= 4) { if constexpr (R - 1 < NI) fa[BUF][R - 1] = *reinterpret_cast<const frag *>(pa + (R - 1) * 32 * ROW_BYTES); }
            else if (R - 4 < NJ) fb[BUF][R - 4] = *reinterpret_cast<const frag *>(pb + (R - 4) * 32 * ROW_BYTES);
        }
    };
    auto dma_one = [&](auto is_b, auto jj, int64_t koff, char *base) {
        constexpr int J = decltype(jj)::value;
        if (W4_ABL & 1) return;
        if constexpr (BNN && decltype(is_b)::value) {
            if constexpr (J >= NPB) return;                                   // the 192-column tile: six pieces of B per wave
            // koff = tile * 128 bytes along K for the K-contiguous layout; here a K-tile is 32 rows of ldb elements
            glds16_s<J * 1024>(ubase_bnn + koff * g.ldb, voff_bnn[J], lds_addr_of(base) + (uint32_t)(wave * DST_PIECE_B_STEP));
        } else if constexpr (ATN && !decltype(is_b)::value) {
            glds16_s<J * 1024>(ubase_atn + koff * g.lda, voff_atn[J], lds_addr_of(base));
        } else {
            if constexpr (decltype(is_b)::value && J >= NPB) return;          // the 192-column tile: six pieces of B per wave
            if constexpr (!decltype(is_b)::value && J >= NPA) return;         // the 192-row tile: six pieces of A per wave
            glds16_s<J * 1024>((decltype(is_b)::value ? ubase_b : ubase_a) + koff, decltype(is_b)::value ? voff_b[J] : voff_a[J],
                               lds_addr_of(base) + (uint32_t)(wave * (decltype(is_b)::value ? DST_PIECE_B_STEP : DST_PIECE_A_STEP)));
        }
    };
    auto mfma_one = [&](auto buf, auto idx, auto step) {
        constexpr int BUF = decltype(buf)::value, I = decltype(idx)::value & 3, J = decltype(idx)::value >> 2;
        if constexpr (DT != MI355_DTYPE_F32 && (J >= NJ || I >= NI)) return;   // (f32 re-indexes idx; it only exists with NJ = NI = 4)
        if constexpr (MX) {
            // first MFMA operand = B fragment (its format in cbsz, its scale first), second = A fragment
            acc[I][J] = mfma_mx<mx_fmt<DTB>::value, mx_fmt<DT>::value, decltype(step)::value>(fb[BUF][J], fa[BUF][I], acc[I][J],
                                                                                             sc_b[J], sc_a[I]);
            return;
        }
        if constexpr ((W4_ABL & 4) != 0) {    // (constexpr: the host pass must not see a 256-bit "v" operand)
            asm volatile("" ::"v"(fb[BUF][J]), "v"(fa[BUF][I]));
            return;
        }
        if constexpr (DT == MI355_DTYPE_F32) {
            // f32: 64 MFMAs per k-step, ordered element-major so that consecutive MFMAs hit different
            // accumulators (group IDX = element IDX/4 of the fragments x accumulator pairs 4*(IDX%4)..+3;
            // an accumulator is revisited after 16 MFMAs).  Four dependent 32x32x2 MFMAs in a row measured
            // ~10 % slower than the issue rate.
            constexpr int E = decltype(idx)::value >> 2;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int pair = (decltype(idx)::value & 3) * 4 + t, i = pair & 3, j = pair >> 2;
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[BUF][j][E], fa[BUF][i][E], acc[i][j], 0, 0, 0);
            }
        } else {
            acc[I][J] = lp<DT>::mfma(fb[BUF][J], fa[BUF][I], acc[I][J]);
        }
    };

    // One k-step, instruction order pinned by hand (sched_barrier(0) after every MFMA group):
    //   MFMA idx, then at most one fragment read of the NEXT k-step (into the other register
    //   buffer) or one DMA piece.  RMASK / DMASK: bit idx set = a read / a DMA follows MFMA idx.
    //   DMA pieces are J0, J0+1, ... in mask order.
    //   MX: STEP = k-step within the K-tile (the scale byte the MFMAs select); SMASK / Q0: scale loads of the next
    //   K-tile, numbered Q0, Q0+1, ... in mask order.
#define W4_STEP_BODY(CUR, NXT, RMASK, DMASK, IS_B, J0, STEP, SMASK, Q0)                              \
    {                                                                                                \
        constexpr unsigned rmask_ = (RMASK), dmask_ = (DMASK), smask_ = (SMASK);                     \
        W4_GROUP(CUR, NXT, 0, IS_B, J0, STEP, Q0)  W4_GROUP(CUR, NXT, 1, IS_B, J0, STEP, Q0)         \
        W4_GROUP(CUR, NXT, 2, IS_B, J0, STEP, Q0)  W4_GROUP(CUR, NXT, 3, IS_B, J0, STEP, Q0)         \
        W4_GROUP(CUR, NXT, 4, IS_B, J0, STEP, Q0)  W4_GROUP(CUR, NXT, 5, IS_B, J0, STEP, Q0)         \
        W4_GROUP(CUR, NXT, 6, IS_B, J0, STEP, Q0)  W4_GROUP(CUR, NXT, 7, IS_B, J0, STEP, Q0)         \
        W4_GROUP(CUR, NXT, 8, IS_B, J0, STEP, Q0)  W4_GROUP(CUR, NXT, 9, IS_B, J0, STEP, Q0)         \
        W4_GROUP(CUR, NXT, 10, IS_B, J0, STEP, Q0) W4_GROUP(CUR, NXT, 11, IS_B, J0, STEP, Q0)        \
        W4_GROUP(CUR, NXT, 12, IS_B, J0, STEP, Q0) W4_GROUP(CUR, NXT, 13, IS_B, J0, STEP, Q0)        \
        W4_GROUP(CUR, NXT, 14, IS_B, J0, STEP, Q0) W4_GROUP(CUR, NXT, 15, IS_B, J0, STEP, Q0)        \
    }
#define W4_GROUP(CUR, NXT, IDX, IS_B, J0, STEP, Q0)                                                  \
    mfma_one(IC<CUR>{}, IC<IDX>{}, IC<STEP>{});                                                      \
    if constexpr ((rmask_ >> IDX) & 1u)                                                              \
        read_one(IC<NXT>{}, IC<__builtin_popcount(rmask_ & ((1u << IDX) - 1u))>{}, rd_a, rd_b);      \
    if constexpr ((smask_ >> IDX) & 1u)                                                              \
        scale_one(IC<Q0 + __builtin_popcount(smask_ & ((1u << IDX) - 1u))>{});                       \
    if constexpr ((dmask_ >> IDX) & 1u)                                                              \
        dma_one(IC<IS_B>{}, IC<J0 + __builtin_popcount(dmask_ & ((1u << IDX) - 1u))>{}, dma_koff, dma_base); \
    __builtin_amdgcn_sched_barrier(0);

    // ---- prologue: (MX: the scales of K-tile 0,) units 0..3 (K-tiles 0 and 1), then the first fragments ------
    scale_one(IC<0>{}); scale_one(IC<1>{}); scale_one(IC<2>{}); scale_one(IC<3>{});
    scale_one(IC<4>{}); scale_one(IC<5>{}); scale_one(IC<6>{}); scale_one(IC<7>{});
    {
        const int64_t k0 = 0, k1 = (int64_t)min(1, nk - 1) * ROW_BYTES;   // nk == 1: K-tile 0 twice, drained at the hand-over
        char *b0 = smem + dst_piece;
#define W4_PRO(IS_B, KOFF, SLOT)                                                                     \
        dma_one(IC<IS_B>{}, IC<0>{}, KOFF, b0 + SLOT * UNIT_BYTES); dma_one(IC<IS_B>{}, IC<1>{}, KOFF, b0 + SLOT * UNIT_BYTES); \
        dma_one(IC<IS_B>{}, IC<2>{}, KOFF, b0 + SLOT * UNIT_BYTES); dma_one(IC<IS_B>{}, IC<3>{}, KOFF, b0 + SLOT * UNIT_BYTES); \
        dma_one(IC<IS_B>{}, IC<4>{}, KOFF, b0 + SLOT * UNIT_BYTES); dma_one(IC<IS_B>{}, IC<5>{}, KOFF, b0 + SLOT * UNIT_BYTES); \
        dma_one(IC<IS_B>{}, IC<6>{}, KOFF, b0 + SLOT * UNIT_BYTES); dma_one(IC<IS_B>{}, IC<7>{}, KOFF, b0 + SLOT * UNIT_BYTES);
        W4_PRO(0, k0, 0) W4_PRO(1, k0, 1) W4_PRO(0, k1, 2) W4_PRO(1, k1, 3)
#undef W4_PRO
    }
    // units 0, 1 landed (this wave's share): units 2, 3 = NPA + NPB pieces may fly
    if constexpr (NPA + NPB == 16) WAIT_VMCNT(16); else if constexpr (NPA + NPB == 14) WAIT_VMCNT(14); else WAIT_VMCNT(12);
    W4_TAKE_SCALES(16)                   // (older than every DMA: landed too)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    {
        // fp4 (MX): lane-half h owns the contiguous half of the row (chunks = MX blocks 4h .. 4h+3), so that its four
        // scale bytes are one word; fp8 MX: k-step d = chunks 4d+h (registers 0-3) and 4d+2+h (registers 4-7);
        // unscaled: the mappings the measured kernels were tuned with
        const int x = F4 ? ((4 * h) ^ f) << 4 : (F8 && !MX) ? ((2 * h) ^ f) << 4 : (h ^ f) << 4;
        const char *rd_a = smem + rowoff_a + (ATN ? h * 4096 : x), *rd_b = smem + UNIT_BYTES + rowoff_b + (BNN16 ? h * 2 * BROW : BNN ? (4 * h) * 1024 : x);
        read_one(IC<0>{}, IC<0>{}, rd_a, rd_b); read_one(IC<0>{}, IC<1>{}, rd_a, rd_b);
        read_one(IC<0>{}, IC<2>{}, rd_a, rd_b); read_one(IC<0>{}, IC<3>{}, rd_a, rd_b);
        read_one(IC<0>{}, IC<4>{}, rd_a, rd_b); read_one(IC<0>{}, IC<5>{}, rd_a, rd_b);
        read_one(IC<0>{}, IC<6>{}, rd_a, rd_b); read_one(IC<0>{}, IC<7>{}, rd_a, rd_b);
        if constexpr (F8) {
            read_one(IC<0>{}, IC<8>{}, rd_a, rd_b); read_one(IC<0>{}, IC<9>{}, rd_a, rd_b);
            read_one(IC<0>{}, IC<10>{}, rd_a, rd_b); read_one(IC<0>{}, IC<11>{}, rd_a, rd_b);
            read_one(IC<0>{}, IC<12>{}, rd_a, rd_b); read_one(IC<0>{}, IC<13>{}, rd_a, rd_b);
            read_one(IC<0>{}, IC<14>{}, rd_a, rd_b); read_one(IC<0>{}, IC<15>{}, rd_a, rd_b);
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    int sa = 0;                          // ring byte offset of unit 2t   (A of K-tile t)
    int sb = UNIT_BYTES;                 // ring byte offset of unit 2t+1 (B of K-tile t)
    auto adv = [](int x, int n) { x += n * UNIT_BYTES; return x >= LDS_BYTES ? x - LDS_BYTES : x; };
    const int x1 = ((F4 ? 4 * h + 1 : 2 + h) ^ f) << 4, x2 = ((F4 ? 4 * h + 2 : 4 + h) ^ f) << 4,
              x3 = ((F4 ? 4 * h + 3 : 6 + h) ^ f) << 4, x0 = ((F4 ? 4 * h : h) ^ f) << 4;
    // B fragment offsets per k-step: same chunks as A for [N][K]; k-rows 8s + 4h (.. +3) for row-major B
    // (16-bit: block rows a = 4s + 2h, + 1: 8 blocks of 256 B per block row)
    const int y0 = BNN16 ? h * 2 * BROW : BNN ? (4 * h) * 1024 : x0, y1 = BNN16 ? (4 + 2 * h) * BROW : BNN ? (8 + 4 * h) * 1024 : x1,
              y2 = BNN16 ? (8 + 2 * h) * BROW : BNN ? (16 + 4 * h) * 1024 : x2, y3 = BNN16 ? (12 + 2 * h) * BROW : BNN ? (24 + 4 * h) * 1024 : x3;

    const int xa0 = ATN ? y0 : x0, xa1 = ATN ? y1 : x1, xa2 = ATN ? y2 : x2, xa3 = ATN ? y3 : x3;   // A stored [K][M]: block rows, as row-major B

    // One K-tile.  ISSUE = 1: the steady state (units 2t+4, 2t+5 are issued, vmcnt(8) at the hand-over).
    // ISSUE = 0: the last two K-tiles of the tile -- there is nothing left to fetch, so no DMA is issued (the first
    // version re-read the last K-tile into dead slots to keep the counts uniform: 2 K-tiles of useless L2 traffic
    // per output tile and a vmcnt(0) stall on them before the epilogue) and the hand-over waits for everything.
#define W4_KTILE(ISSUE, SC)                                                                                  \
    {                                                                                                       \
        const int sa1 = adv(sa, 2), sb1 = adv(sb, 2);     /* units 2t+2, 2t+3 (K-tile t+1) */               \
        const int s4 = adv(sa, 4);                        /* unit 2t+4 -> slot of unit 2t-1 */               \
        const int s5 = sa;                                /* unit 2t+5 -> slot of unit 2t   */               \
        const int64_t dma_koff = (int64_t)(t + 2) * ROW_BYTES;                                              \
        const char *rd_a, *rd_b;                                                                            \
        char *dma_base;                                                                                     \
        /* k-step 0: reads of step 1 after MFMA 0-7, unit 2t+4 pieces 0-3 after MFMA 9,11,13,15 */          \
        rd_a = smem + sa + rowoff_a + xa1; rd_b = smem + sb + rowoff_b + y1; dma_base = smem + s4 + dst_piece; \
        /* (MX: the 8 scale loads of K-tile t+1 after MFMA 8,10,12,14 of k-steps 0 and 1) */                \
        W4_STEP_BODY(0, 1, 0x00FFu, (ISSUE) ? 0xAA00u : 0u, 0, 0, 0, (MX && (SC)) ? 0x5500u : 0u, 0)         \
        /* k-step 1: reads of step 2, unit 2t+4 pieces 4-7 */                                               \
        rd_a = smem + sa + rowoff_a + xa2; rd_b = smem + sb + rowoff_b + y2;                                 \
        W4_STEP_BODY(1, 0, 0x00FFu, (ISSUE) ? 0xAA00u : 0u, 0, 4, 1, (MX && (SC)) ? 0x5500u : 0u, 4)         \
        /* k-step 2: reads of step 3, no DMA; then the K-tile hand-over */                                  \
        rd_a = smem + sa + rowoff_a + xa3; rd_b = smem + sb + rowoff_b + y3;                                 \
        W4_STEP_BODY(0, 1, 0x00FFu, 0x0000u, 0, 0, 2, 0u, 0)                                                \
        if (!(W4_ABL & 8)) {                              /* dev ablation 8: no hand-over (timing only, races) */ \
        /* my share of K-tile t+1 landed; unit 2t+4 (and, MX, the 8 scale loads issued among it) may fly */ \
        if (ISSUE) { if constexpr (MX) WAIT_VMCNT(16); else if constexpr (NPA == 8) WAIT_VMCNT(8); else WAIT_VMCNT(6); } else WAIT_VMCNT(0); \
        WAIT_LGKM0();                                     /* my reads of K-tile t are complete */           \
        __builtin_amdgcn_s_barrier();                     /* BAR_t */                                        \
        }                                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        /* k-step 3: reads of step 0 of K-tile t+1 after even MFMAs, unit 2t+5 pieces 0-7 after odd ones */ \
        rd_a = smem + sa1 + rowoff_a + xa0; rd_b = smem + sb1 + rowoff_b + y0; dma_base = smem + s5 + dst_piece; \
        W4_STEP_BODY(1, 0, 0x5555u, (ISSUE) ? 0xAAAAu : 0u, 1, 0, 3, 0u, 0)                                  \
        if constexpr (MX && (SC)) { if (ISSUE) { W4_TAKE_SCALES(8) } else { W4_TAKE_SCALES(0) } }           \
        sa = sa1;                                                                                           \
        sb = sb1;                                                                                           \
    }
    // fp8 K-tile: two k-steps of 64 (d = 0, 1), 16 MFMAs of 64 cycles each; register buffer 0 always holds the
    // fragments of d = 0, buffer 1 those of d = 1.  Same hand-over point as above (three quarters through):
    //     d=0, MFMA 0-15 : one read of frags(t, d=1) after each; DMA unit 2t+4 pieces 0-7 after the odd ones
    //     d=1, MFMA 0-7  : nothing else
    //          vmcnt(8), lgkmcnt(0), s_barrier (BAR_t)
    //     d=1, MFMA 8-15 : two reads of frags(t+1, d=0) and one DMA piece of unit 2t+5 after each
    //   MX: the 8 scale loads of K-tile t+1 follow the even MFMAs of d=0 (SQ = load number, -1 = none).
#define W8_G(CUR, NXT, IDX, NR, R0, DM, IS_B, J, SQ)                                                        \
    mfma_one(IC<CUR>{}, IC<IDX>{}, IC<2 * CUR>{});        /* buffer number == k-step d; MX scale byte 2d */   \
    if constexpr ((NR) >= 1) read_one(IC<NXT>{}, IC<(R0)>{}, rd_a, rd_b);                                    \
    if constexpr ((NR) >= 2) read_one(IC<NXT>{}, IC<(R0) + 1>{}, rd_a, rd_b);                                \
    if constexpr ((SQ) >= 0) scale_one(IC<((SQ) >= 0 ? (SQ) : 0)>{});                                        \
    if constexpr (DM) dma_one(IC<IS_B>{}, IC<J>{}, dma_koff, dma_base);                                      \
    __builtin_amdgcn_sched_barrier(0);
#define W8_SQ(SC, Q) ((MX && (SC)) ? (Q) : -1)
#define W8_KTILE(ISSUE, SC)                                                                                  \
    {                                                                                                       \
        const int sa1 = adv(sa, 2), sb1 = adv(sb, 2);                                                       \
        const int s4 = adv(sa, 4);                                                                          \
        const int s5 = sa;                                                                                  \
        const int64_t dma_koff = (int64_t)(t + 2) * ROW_BYTES;                                              \
        const char *rd_a, *rd_b;                                                                            \
        char *dma_base;                                                                                     \
        rd_a = smem + sa + rowoff_a + z1; rd_b = smem + sb + rowoff_b + z1; dma_base = smem + s4 + dst_piece; \
        W8_G(0, 1, 0, 1, 0, 0, 0, 0, W8_SQ(SC, 0))     W8_G(0, 1, 1, 1, 1, (ISSUE), 0, 0, -1)                 \
        W8_G(0, 1, 2, 1, 2, 0, 0, 0, W8_SQ(SC, 1))     W8_G(0, 1, 3, 1, 3, (ISSUE), 0, 1, -1)                 \
        W8_G(0, 1, 4, 1, 4, 0, 0, 0, W8_SQ(SC, 2))     W8_G(0, 1, 5, 1, 5, (ISSUE), 0, 2, -1)                 \
        W8_G(0, 1, 6, 1, 6, 0, 0, 0, W8_SQ(SC, 3))     W8_G(0, 1, 7, 1, 7, (ISSUE), 0, 3, -1)                 \
        W8_G(0, 1, 8, 1, 8, 0, 0, 0, W8_SQ(SC, 4))     W8_G(0, 1, 9, 1, 9, (ISSUE), 0, 4, -1)                 \
        W8_G(0, 1, 10, 1, 10, 0, 0, 0, W8_SQ(SC, 5))   W8_G(0, 1, 11, 1, 11, (ISSUE), 0, 5, -1)               \
        W8_G(0, 1, 12, 1, 12, 0, 0, 0, W8_SQ(SC, 6))   W8_G(0, 1, 13, 1, 13, (ISSUE), 0, 6, -1)               \
        W8_G(0, 1, 14, 1, 14, 0, 0, 0, W8_SQ(SC, 7))   W8_G(0, 1, 15, 1, 15, (ISSUE), 0, 7, -1)               \
        W8_G(1, 0, 0, 0, 0, 0, 0, 0, -1) W8_G(1, 0, 1, 0, 0, 0, 0, 0, -1) W8_G(1, 0, 2, 0, 0, 0, 0, 0, -1) W8_G(1, 0, 3, 0, 0, 0, 0, 0, -1) \
        W8_G(1, 0, 4, 0, 0, 0, 0, 0, -1) W8_G(1, 0, 5, 0, 0, 0, 0, 0, -1) W8_G(1, 0, 6, 0, 0, 0, 0, 0, -1) W8_G(1, 0, 7, 0, 0, 0, 0, 0, -1) \
        if (!(W4_ABL & 8)) {                                                                                \
        if (ISSUE) { if constexpr (MX) WAIT_VMCNT(16); else WAIT_VMCNT(8); } else WAIT_VMCNT(0);            \
        WAIT_LGKM0();                                                                                       \
        __builtin_amdgcn_s_barrier();                                                                       \
        }                                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        rd_a = smem + sa1 + rowoff_a + z0; rd_b = smem + sb1 + rowoff_b + z0; dma_base = smem + s5 + dst_piece; \
        W8_G(1, 0, 8, 2, 0, (ISSUE), 1, 0, -1)      W8_G(1, 0, 9, 2, 2, (ISSUE), 1, 1, -1)                    \
        W8_G(1, 0, 10, 2, 4, (ISSUE), 1, 2, -1)     W8_G(1, 0, 11, 2, 6, (ISSUE), 1, 3, -1)                   \
        W8_G(1, 0, 12, 2, 8, (ISSUE), 1, 4, -1)     W8_G(1, 0, 13, 2, 10, (ISSUE), 1, 5, -1)                  \
        W8_G(1, 0, 14, 2, 12, (ISSUE), 1, 6, -1)    W8_G(1, 0, 15, 2, 14, (ISSUE), 1, 7, -1)                  \
        if constexpr (MX && (SC)) { if (ISSUE) { W4_TAKE_SCALES(8) } else { W4_TAKE_SCALES(0) } }           \
        sa = sa1;                                                                                           \
        sb = sb1;                                                                                           \
    }
    // fp8: first chunk of this lane-half for k-steps d = 0, 1 (MX: chunk 4d + h, its partner 4d + 2 + h via hd)
    const int z0 = ((MX ? h : 2 * h) ^ f) << 4, z1 = ((MX ? 4 + h : 4 + 2 * h) ^ f) << 4;
    W4_STAMP(1);
    int t = 0;
    // MX: every K-tile but the last also fetches the next one's scales (SC), so the DMA-free tail splits in two
    if constexpr (F8 && MX) {
        for (; t + 2 < nk; ++t) W8_KTILE(1, 1)
        for (; t + 1 < nk; ++t) W8_KTILE(0, 1)
        for (; t < nk; ++t) W8_KTILE(0, 0)
    } else if constexpr (F8) {
        for (; t + 2 < nk; ++t) W8_KTILE(1, 0)
        for (; t < nk; ++t) W8_KTILE(0, 0)
    } else if constexpr (MX) {
        for (; t + 2 < nk; ++t) W4_KTILE(1, 1)
        for (; t + 1 < nk; ++t) W4_KTILE(0, 1)
        for (; t < nk; ++t) W4_KTILE(0, 0)
    } else {
        for (; t + 2 < nk; ++t) W4_KTILE(1, 0)
        for (; t < nk; ++t) W4_KTILE(0, 0)
    }
#undef W4_KTILE
#undef W8_KTILE
#undef W8_G
#undef W8_SQ
#undef W4_TAKE_SCALES
    W4_STAMP(2);
#undef W4_STEP_BODY
#undef W4_GROUP
    // nothing is in flight here: the last hand-over waited for vmcnt(0) and no DMA was issued after it

    // ---- epilogue ----------------------------------------------------------------------------------
    // With the operands swapped in the MFMA (first = B fragment), lane (l31, h) of a wave holds, for
    // every 32-row block i, row l31 of the block and the column groups n = j*32 + 8q + 4h .. +3.
    // Stored straight from registers that is 8 B per lane on 32 different rows per instruction: every
    // 128-byte line of C is assembled from 16 partial writes (measured: ~39k cycles of fixed cost per
    // output tile).  Instead each wave transposes its 128 x 128 block through its own LDS scratch, 32
    // rows at a time, and writes whole rows: 16 B per lane, 256 (16-bit) / 512 (f32) contiguous bytes
    // per row, every line written once.  Row pitch +16 B keeps the b128 accesses aligned and the
    // writes at most 2-way bank conflicted.
    char *__restrict__ C = static_cast<char *>(g.c);
    constexpr int CSZ = (DT_C == MI355_DTYPE_F32) ? 4 : 2;
    const int64_t cbase = batch * g.stride_c;
    {
        constexpr int RS = 128 * CSZ + 16;                 // staged row pitch in bytes
        constexpr int STAGE = 32 * RS;                     // per-wave scratch: 8.5 KiB (16-bit) / 16.5 KiB (f32)
        constexpr int LPR = 128 * CSZ / 16;                // lanes per output row: 16 / 32
        constexpr int RPI = 64 / LPR;                      // rows per store instruction: 4 / 2
        __builtin_amdgcn_s_barrier();                      // every wave's tail DMA has landed; LDS is free
        char *stage = smem + wave * ((STAGE + 1023) & ~1023);
        char *wr = stage + l31 * RS + 4 * h * CSZ;
        const char *rd = stage + (lane / LPR) * RS + (lane % LPR) * 16;
        char *crow = C + (cbase + (m0 + wm * (NI * 32) + lane / LPR) * g.ldc + n0 + wn * (NJ * 32)) * CSZ + (lane % LPR) * 16;
        const int64_t cstep = (int64_t)RPI * g.ldc * CSZ;
        // edge tiles: rows >= M and columns >= N are not stored (a 16-byte piece straddling N is written element-wise)
        constexpr int EPP = 16 / CSZ;                                     // elements per 16-byte piece
        const int64_t row0 = m0 + wm * (NI * 32) + lane / LPR;            // + i*32 + it*RPI
        const int64_t col0 = n0 + wn * (NJ * 32) + (lane % LPR) * EPP;
        // (the 192-column tile: a wave's rows are NJ * 32 columns wide -- the lanes of a row beyond that have nothing to store)
        const bool lane_live = NJ == 4 || (int)(lane % LPR) * EPP < NJ * 32;
        const int ncols = lane_live ? (int)max((int64_t)0, min((int64_t)EPP, g.n - col0)) : 0;   // valid elements of my piece
        // C rows that do not start on 16-byte boundaries (N = 50257 bf16 logits, a view at an odd column): every piece goes
        // the element-wise way of the edge tiles -- eight 2-byte stores per lane instead of one 16-byte store, which the L2
        // merges into the same lines (8191 x 8191 x 8192: see profiles/r04_ragged_probe.txt).
        const bool cvec = ((((uint64_t)g.ldc * CSZ) | ((uint64_t)g.stride_c * CSZ) | reinterpret_cast<uintptr_t>(g.c)) & 15u) == 0;
        const bool interior = cvec && (m0 + BMT <= g.m) && (n0 + BNT <= g.n);   // wave-uniform fast path
        // D = A * B + c_in (f32 C only, the C operand of cmma::execute): the 32 / RPI pieces of c_in that this lane will
        // add in block i are fetched before the block's accumulators are staged, so one memory latency per block hides
        // behind the LDS transposition.  c_in has C's layout and may be C itself.
        const char *cin = nullptr;
        if constexpr (DT_C == MI355_DTYPE_F32) cin = static_cast<const char *>(g.c_in);
        const int64_t cin_off = crow - C;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            f32x4 pre[DT_C == MI355_DTYPE_F32 ? 32 / RPI : 1];
            if constexpr (DT_C == MI355_DTYPE_F32) {
                if (cin) {
                    const char *csrc = cin + cin_off + (int64_t)i * 32 * g.ldc * CSZ;
#pragma unroll
                    for (int it = 0; it < 32 / RPI; ++it) {
                        if ((interior && lane_live) || (cvec && row0 + i * 32 + it * RPI < g.m && ncols == EPP))
                            pre[it] = *reinterpret_cast<const f32x4 *>(csrc + it * cstep);
                        else
                            pre[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    char *d = wr + (j * 32 + 8 * q) * CSZ;
                    if constexpr (DT_C == MI355_DTYPE_F32) {
                        f32x4 v = {acc[i][j][4 * q + 0], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                        *reinterpret_cast<f32x4 *>(d) = v;
                    } else if constexpr (DT_C == MI355_DTYPE_BF16) {
                        bf16x4 v = {(__bf16)acc[i][j][4 * q + 0], (__bf16)acc[i][j][4 * q + 1], (__bf16)acc[i][j][4 * q + 2],
                                    (__bf16)acc[i][j][4 * q + 3]};
                        *reinterpret_cast<bf16x4 *>(d) = v;
                    } else {
                        f16x4 v = {(_Float16)acc[i][j][4 * q + 0], (_Float16)acc[i][j][4 * q + 1], (_Float16)acc[i][j][4 * q + 2],
                                   (_Float16)acc[i][j][4 * q + 3]};
                        *reinterpret_cast<f16x4 *>(d) = v;
                    }
                }
            WAIT_LGKM0();                                  // same-wave hand-over: DS ops of one wave execute in order
            char *cdst = crow + (int64_t)i * 32 * g.ldc * CSZ;
#pragma unroll
            for (int it = 0; it < 32 / RPI; ++it) {
                u32x4 v = *reinterpret_cast<const u32x4 *>(rd + it * RPI * RS);
                if (NJ != 4 && !lane_live) continue;
                if (!interior) {
                    if (row0 + i * 32 + it * RPI >= g.m || ncols <= 0) continue;
                    if (ncols < EPP || !cvec) {
#pragma unroll
                        for (int e = 0; e < EPP; ++e) {                    // static indices only (guide rule 20)
                            if (e >= ncols) break;
                            if constexpr (CSZ == 4) {
                                float x = __uint_as_float(v[e]);
                                if (cin) x += reinterpret_cast<const float *>(cin + cin_off + (int64_t)i * 32 * g.ldc * CSZ + it * cstep)[e];
                                reinterpret_cast<float *>(cdst + it * cstep)[e] = x;
                            } else
                                reinterpret_cast<uint16_t *>(cdst + it * cstep)[e] = (uint16_t)(v[e >> 1] >> ((e & 1) * 16));
                        }
                        continue;
                    }
                }
                if constexpr (DT_C == MI355_DTYPE_F32) {
                    if (cin) {
                        const f32x4 sum = __builtin_bit_cast(f32x4, v) + pre[it];
                        v = __builtin_bit_cast(u32x4, sum);
                    }
                }
                if ((W4_ABL & 16) && g.m > 1) continue;       // dev ablation 16: no C stores (timing only)
#if W4_NT_C
                __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(cdst + it * cstep));
#else
                *reinterpret_cast<u32x4 *>(cdst + it * cstep) = v;
#endif
            }
            __builtin_amdgcn_sched_barrier(0);             // keep the accumulator reads of block i+1 below this point
        }
    }
#ifdef W4_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    W4_STAMP(3);
#endif
}

template <int DT, int DT_C, bool BNN = false, int DTB = DT, bool MX = false, bool ATN = false, int NJ = 4, int NI = 4>
void launch(mi355_ctx *ctx, hipStream_t s, const gemm_args &g, uint32_t batch)
{
    lds_opt_in(ctx, reinterpret_cast<const void *>(gemm_lp256w4_kernel<DT, DT_C, BNN, DTB, MX, ATN, NJ, NI>), LDS_BYTES);
    hipLaunchKernelGGL((gemm_lp256w4_kernel<DT, DT_C, BNN, DTB, MX, ATN, NJ, NI>), dim3(g.tiles_m * g.tiles_n, batch), dim3(256), LDS_BYTES, s, g);
}

}  // namespace

#ifdef W4_TRACE
extern "C" __attribute__((visibility("default"))) int mi355_dev_w4_trace(unsigned long long *host_out)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(w4_trace_buf), sizeof(unsigned long long) * 4096 * 8);
}
#endif

namespace mi355 {

bool gemm_lp256w4_supports(const mi355_gemm_desc &d, const void *a, const void *b, const void *c)
{
    const bool f8 = d.dtype_ab == MI355_DTYPE_F8E4M3 || d.dtype_ab == MI355_DTYPE_F8E5M2;
    if (d.dtype_ab != MI355_DTYPE_BF16 && d.dtype_ab != MI355_DTYPE_F16 && d.dtype_ab != MI355_DTYPE_F32 && !f8) return false;
    if (f8) {
        if (d.dtype_c != MI355_DTYPE_F32 && d.dtype_c != MI355_DTYPE_BF16 && d.dtype_c != MI355_DTYPE_F16) return false;
    } else if (d.dtype_c != MI355_DTYPE_F32 && d.dtype_c != d.dtype_ab) return false;
    // A stored [K][M]: 16-bit operands, together with a row-major B, fetched 8 rows of C (16 bytes of a k-row) at a time
    if (d.trans_a && ((d.dtype_ab != MI355_DTYPE_BF16 && d.dtype_ab != MI355_DTYPE_F16) || d.trans_b || d.m < 8 || (d.m & 7))) return false;
    if (!d.trans_b && f8) return false;                                     // row-major B: f32 and 16-bit operands
    const int64_t esz = f8 ? 1 : d.dtype_ab == MI355_DTYPE_F32 ? 4 : 2;
    const int64_t BK = ROW_BYTES / esz;
    if (d.k < BK || d.k % BK != 0) return false;
    // (C rows off the 16-byte grid are taken: the epilogue then stores element-wise; the persistent forms still decline them)
    const int64_t csz = d.dtype_c == MI355_DTYPE_F32 ? 4 : 2;
    if (reinterpret_cast<uintptr_t>(c) & (uintptr_t)(csz - 1)) return false;
    if (d.m < 1 || d.n < 1) return false;
    // row-major B is fetched 16 bytes of a row at a time: 4 (f32) / 8 (16-bit) columns
    if (!d.trans_b && (d.n < 16 / esz || (d.n & (16 / esz - 1)))) return false;
    const int64_t amask = 16 / esz - 1;                               // operand rows must be 16-byte aligned
    if ((d.lda & amask) || (d.ldb & amask) || (d.stride_a & amask) || (d.stride_b & amask)) return false;
    if ((reinterpret_cast<uintptr_t>(a) & 15u) || (reinterpret_cast<uintptr_t>(b) & 15u)) return false;
    if (d.batch > 65535) return false;
    const int64_t tiles = ((d.m + BM - 1) / BM) * ((d.n + BN - 1) / BN);
    if (tiles * std::max<int64_t>(d.batch, 1) > 0x7FFFFFFF) return false;   // the XCD remap runs over the (batch, tile) sequence in 32 bits
    if ((int64_t)BM * std::max(d.lda, d.ldb) * esz >= (1ll << 32)) return false;   // per-lane DMA offsets are 32-bit
    return true;
}

int32_t launch_gemm_lp256w4(mi355_ctx *ctx, hipStream_t s, const mi355_gemm_desc &d, const void *a, const void *b,
                            void *c, const void *c_in)
{
    if (!gemm_lp256w4_supports(d, a, b, c))
        return fail(ctx, MI355_E_UNSUPPORTED, "lp256w4 GEMM: shape/layout not supported by this kernel");
    if (c_in && (d.dtype_c != MI355_DTYPE_F32 || (reinterpret_cast<uintptr_t>(c_in) & 15u)))
        return fail(ctx, MI355_E_UNSUPPORTED, "lp256w4 GEMM: the in-kernel C operand needs f32 output and 16-byte alignment");
    gemm_args g{};
    g.a = a; g.b = b; g.c = c;
    g.c_in = c_in;
    g.m = d.m; g.n = d.n; g.k = d.k;
    g.lda = d.lda; g.ldb = d.ldb; g.ldc = d.ldc;
    g.stride_a = d.stride_a; g.stride_b = d.stride_b; g.stride_c = d.stride_c;
    g.tiles_m = (uint32_t)((d.m + BM - 1) / BM);
    g.tiles_n = (uint32_t)((d.n + BN - 1) / BN);
    g.group_m = W4_GROUP_M;
    const uint32_t batch = (uint32_t)d.batch;
    if (d.dtype_ab == MI355_DTYPE_F8E4M3) {
        if (d.dtype_c == MI355_DTYPE_F32) launch<MI355_DTYPE_F8E4M3, MI355_DTYPE_F32>(ctx, s, g, batch);
        else if (d.dtype_c == MI355_DTYPE_BF16) launch<MI355_DTYPE_F8E4M3, MI355_DTYPE_BF16>(ctx, s, g, batch);
        else launch<MI355_DTYPE_F8E4M3, MI355_DTYPE_F16>(ctx, s, g, batch);
    } else if (d.dtype_ab == MI355_DTYPE_F8E5M2) {
        if (d.dtype_c == MI355_DTYPE_F32) launch<MI355_DTYPE_F8E5M2, MI355_DTYPE_F32>(ctx, s, g, batch);
        else if (d.dtype_c == MI355_DTYPE_BF16) launch<MI355_DTYPE_F8E5M2, MI355_DTYPE_BF16>(ctx, s, g, batch);
        else launch<MI355_DTYPE_F8E5M2, MI355_DTYPE_F16>(ctx, s, g, batch);
    } else if (d.dtype_ab == MI355_DTYPE_F32) {
        if (d.trans_b) launch<MI355_DTYPE_F32, MI355_DTYPE_F32, false>(ctx, s, g, batch);
        else launch<MI355_DTYPE_F32, MI355_DTYPE_F32, true>(ctx, s, g, batch);
    } else if (d.dtype_ab == MI355_DTYPE_BF16) {
        if (d.trans_a) {
            if (d.dtype_c == MI355_DTYPE_F32) launch<MI355_DTYPE_BF16, MI355_DTYPE_F32, true, MI355_DTYPE_BF16, false, true>(ctx, s, g, batch);
            else launch<MI355_DTYPE_BF16, MI355_DTYPE_BF16, true, MI355_DTYPE_BF16, false, true>(ctx, s, g, batch);
        } else if (d.trans_b) {
            if (d.dtype_c == MI355_DTYPE_F32) launch<MI355_DTYPE_BF16, MI355_DTYPE_F32>(ctx, s, g, batch);
            else launch<MI355_DTYPE_BF16, MI355_DTYPE_BF16>(ctx, s, g, batch);
        } else {
            if (d.dtype_c == MI355_DTYPE_F32) launch<MI355_DTYPE_BF16, MI355_DTYPE_F32, true>(ctx, s, g, batch);
            else launch<MI355_DTYPE_BF16, MI355_DTYPE_BF16, true>(ctx, s, g, batch);
        }
    } else {
        if (d.trans_a) {
            if (d.dtype_c == MI355_DTYPE_F32) launch<MI355_DTYPE_F16, MI355_DTYPE_F32, true, MI355_DTYPE_F16, false, true>(ctx, s, g, batch);
            else launch<MI355_DTYPE_F16, MI355_DTYPE_F16, true, MI355_DTYPE_F16, false, true>(ctx, s, g, batch);
        } else if (d.trans_b) {
            if (d.dtype_c == MI355_DTYPE_F32) launch<MI355_DTYPE_F16, MI355_DTYPE_F32>(ctx, s, g, batch);
            else launch<MI355_DTYPE_F16, MI355_DTYPE_F16>(ctx, s, g, batch);
        } else {
            if (d.dtype_c == MI355_DTYPE_F32) launch<MI355_DTYPE_F16, MI355_DTYPE_F32, true>(ctx, s, g, batch);
            else launch<MI355_DTYPE_F16, MI355_DTYPE_F16, true>(ctx, s, g, batch);
        }
    }
    check_launch(ctx, "mi355_gemm(lp256w4)");
    return MI355_OK;
}

// ---- the 256 x 192 tile (NJ = 3): [N][K] 16-bit operands ---------------------------------------------------------------------
bool gemm_lp256x192_supports(const mi355_gemm_desc &d, const void *a, const void *b, const void *c)
{
    if (d.dtype_ab != MI355_DTYPE_BF16 && d.dtype_ab != MI355_DTYPE_F16) return false;
    if (d.trans_a) return false;                      // (B: [N][K], or row-major [K][N] through the transposing-read image, as the square tile)
    return gemm_lp256w4_supports(d, a, b, c);
}

int32_t launch_gemm_lp256x192(mi355_ctx *ctx, hipStream_t s, const mi355_gemm_desc &d, const void *a, const void *b, void *c, int tile_rows)
{
    if (!gemm_lp256x192_supports(d, a, b, c) || (tile_rows != 256 && tile_rows != 192))
        return fail(ctx, MI355_E_UNSUPPORTED, "lp256x192 GEMM: shape/layout not supported by this kernel");
    gemm_args g{};
    g.a = a; g.b = b; g.c = c;
    g.m = d.m; g.n = d.n; g.k = d.k;
    g.lda = d.lda; g.ldb = d.ldb; g.ldc = d.ldc;
    g.stride_a = d.stride_a; g.stride_b = d.stride_b; g.stride_c = d.stride_c;
    g.tiles_m = (uint32_t)((d.m + tile_rows - 1) / tile_rows);
    g.tiles_n = (uint32_t)((d.n + 191) / 192);
    g.group_m = W4_GROUP_M;
    const uint32_t batch = (uint32_t)d.batch;
    constexpr int BF = MI355_DTYPE_BF16, HF = MI355_DTYPE_F16, CF = MI355_DTYPE_F32;
#define X192_LAUNCH(NN, NI_)                                                                                   \
    do {                                                                                                       \
        if (d.dtype_ab == BF) {                                                                                \
            if (d.dtype_c == CF) launch<BF, CF, NN, BF, false, false, 3, NI_>(ctx, s, g, batch);                \
            else launch<BF, BF, NN, BF, false, false, 3, NI_>(ctx, s, g, batch);                                \
        } else {                                                                                               \
            if (d.dtype_c == CF) launch<HF, CF, NN, HF, false, false, 3, NI_>(ctx, s, g, batch);                \
            else launch<HF, HF, NN, HF, false, false, 3, NI_>(ctx, s, g, batch);                                \
        }                                                                                                      \
    } while (0)
    if (d.trans_b) { if (tile_rows == 192) X192_LAUNCH(false, 3); else X192_LAUNCH(false, 4); }
    else { if (tile_rows == 192) X192_LAUNCH(true, 3); else X192_LAUNCH(true, 4); }
#undef X192_LAUNCH
    check_launch(ctx, "mi355_gemm(lp256x192)");
    return MI355_OK;
}

// ---- block-scaled (MX) form ---------------------------------------------------------------------------------------------
bool gemm_lp256w4_mx_supports(const mi355_gemm_scaled_desc &d, const void *a, const void *b, const void *c)
{
    const bool f4 = d.dtype_a == MI355_DTYPE_F4E2M1X2;
    if (f4 ? d.dtype_b != MI355_DTYPE_F4E2M1X2 : !(is_fp8(d.dtype_a) && is_fp8(d.dtype_b))) return false;
    if (d.dtype_c != MI355_DTYPE_F32 && d.dtype_c != MI355_DTYPE_BF16) return false;           // (f16 output: generic kernel)
    if (d.block != 32) return false;
    const int64_t bk = f4 ? 256 : 128;                                    // k-values per 128-byte K-tile row
    if (d.k < bk || d.k % bk != 0) return false;
    const int64_t epb = f4 ? 2 : 1;                                       // elements per byte
    if ((d.lda % (16 * epb)) || (d.ldb % (16 * epb)) || (d.stride_a % (16 * epb)) || (d.stride_b % (16 * epb))) return false;
    if ((reinterpret_cast<uintptr_t>(a) & 15u) || (reinterpret_cast<uintptr_t>(b) & 15u)) return false;
    const int64_t csz = d.dtype_c == MI355_DTYPE_F32 ? 4 : 2;
    if (((d.ldc * csz) & 15) || ((d.stride_c * csz) & 15) || (reinterpret_cast<uintptr_t>(c) & 15u)) return false;
    if (d.m < 1 || d.n < 1 || d.batch > 65535) return false;
    if (((d.m + BM - 1) / BM) * ((d.n + BN - 1) / BN) * std::max<int64_t>(d.batch, 1) > 0x7FFFFFFF) return false;
    if ((int64_t)BM * std::max(d.lda, d.ldb) / epb >= (1ll << 32)) return false;
    return true;
}

int32_t launch_gemm_lp256w4_mx(mi355_ctx *ctx, hipStream_t s, const mi355_gemm_scaled_desc &d, const void *a, const void *sa_t,
                               int64_t stride_sa_t, const void *b, const void *sb_t, int64_t stride_sb_t, void *c)
{
    if (!gemm_lp256w4_mx_supports(d, a, b, c))
        return fail(ctx, MI355_E_UNSUPPORTED, "lp256w4 block-scaled GEMM: shape/layout not supported by this kernel");
    const int64_t epb = d.dtype_a == MI355_DTYPE_F4E2M1X2 ? 2 : 1;        // fp4: the kernel works on the byte matrix
    gemm_args g{};
    g.a = a; g.b = b; g.c = c;
    g.m = d.m; g.n = d.n; g.k = d.k / epb;
    g.lda = d.lda / epb; g.ldb = d.ldb / epb; g.ldc = d.ldc;
    g.stride_a = d.stride_a / epb; g.stride_b = d.stride_b / epb; g.stride_c = d.stride_c;
    g.tiles_m = (uint32_t)((d.m + BM - 1) / BM);
    g.tiles_n = (uint32_t)((d.n + BN - 1) / BN);
    g.group_m = W4_GROUP_M;
    g.sa = sa_t; g.sb = sb_t; g.stride_sa = stride_sa_t; g.stride_sb = stride_sb_t;
    const uint32_t batch = (uint32_t)d.batch;
    const bool f32c = d.dtype_c == MI355_DTYPE_F32;
    constexpr int E4 = MI355_DTYPE_F8E4M3, E5 = MI355_DTYPE_F8E5M2, F4 = MI355_DTYPE_F4E2M1X2, CF = MI355_DTYPE_F32, CB = MI355_DTYPE_BF16;
    if (d.dtype_a == F4) {
        if (f32c) launch<F4, CF, false, F4, true>(ctx, s, g, batch); else launch<F4, CB, false, F4, true>(ctx, s, g, batch);
    } else if (d.dtype_a == E4 && d.dtype_b == E4) {
        if (f32c) launch<E4, CF, false, E4, true>(ctx, s, g, batch); else launch<E4, CB, false, E4, true>(ctx, s, g, batch);
    } else if (d.dtype_a == E5 && d.dtype_b == E5) {
        if (f32c) launch<E5, CF, false, E5, true>(ctx, s, g, batch); else launch<E5, CB, false, E5, true>(ctx, s, g, batch);
    } else if (d.dtype_a == E4) {
        if (f32c) launch<E4, CF, false, E5, true>(ctx, s, g, batch); else launch<E4, CB, false, E5, true>(ctx, s, g, batch);
    } else {
        if (f32c) launch<E5, CF, false, E4, true>(ctx, s, g, batch); else launch<E5, CB, false, E4, true>(ctx, s, g, batch);
    }
    check_launch(ctx, "mi355_gemm_scaled(lp256w4)");
    return MI355_OK;
}

}  // namespace mi355
