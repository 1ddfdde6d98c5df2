// gemm_lp128.hip -- bf16 / f16 GEMM, 128x128x64 workgroup tile, v_mfma_f32_32x32x16_{bf16,f16}.
//
// Roofline: MFMA bf16/f16, ~2.5 PFLOP/s dense.  This is the medium-tile kernel: 4 waves (2x2),
// each a 64x64 output (2x2 MFMA tiles of 32x32, 64 accumulator registers), 2 workgroups per CU.
// It serves shapes too small to fill the chip with 256x256 tiles and is the validated base of
// the staging scheme the 256x256 kernels (gemm_lp256w4.hip and its persistent forms) reuse:
//
//  * HBM -> LDS by LDS-DMA (`global_load_lds_dwordx4`, 16 B per lane, no VGPR round trip;
//    cdna_hip_programming.md section 5).  One wave instruction fills 1 KiB = 8 tile rows of
//    64 x 16-bit.  The LDS image is lane-linear, so the bank swizzle is applied to the per-lane
//    SOURCE address and, identically, to the fragment read (guide rule 21):
//        physical 16-byte chunk = logical chunk ^ ((row >> 1) & 7)      (within a 128-byte row)
//    With MFMA 32x32x16 fragment reads (lane -> row, lane-half -> chunk) every 16-lane group of a
//    ds_read_b128 then touches all 16 distinct 16-byte slots of the 256-byte bank row: conflict
//    free.  The source permutation stays inside one 128-byte line, so coalescing is unchanged.
//  * double-buffered LDS, the next tile's DMA issued before the current tile's MFMAs, one
//    vmcnt(0)+barrier per K-tile (guide T3+T4 "minimum 2-phase" recipe).
//  * both operands K-contiguous: A row-major [M][K], B as [N][K] (trans_b = 1, the cmma tests'
//    ColMajor-B form, runtime_tests/cmma.rs:23) -- or, BNN (16-bit operands, round 3), B row-major [K][N] as
//    TensorHandle::new_contiguous lays a rhs out: the B tile (64 k-rows x 128 n) is then staged as 64 blocks of
//    [4 k][32 n] = 256 contiguous bytes (block (a, b) at (4a + b) * 256; one DMA piece = one block row a = 4 k-rows x 256
//    contiguous bytes of global memory) and a B fragment is two ds_read_b64_tr_b16 (gemm_lp256w4.hip, "BNN, bf16 / f16").
//  * MFMA operands swapped (first = B fragment, second = A fragment): each lane then owns 4
//    consecutive N-columns of one C row per register quad -> 16-byte (f32) / 8-byte (16-bit) stores.
#include <algorithm>
#include <type_traits>

#include "gemm_common.hpp"

using namespace mi355;

namespace {

constexpr int BM = 128, BN = 128;
constexpr int ROW_BYTES = 128;                 // one K-tile row = one 128-byte line: 64 x 16-bit or 128 x fp8 k-values
constexpr int TILE_BYTES = BM * ROW_BYTES;     // 16 KiB per operand per stage (the B tile always; the A tile for MI = 2)
// MI = 32-row blocks per wave along M.  2: the 128 x 128 tile (waves 2 x 2, 64 x 64 each).  4 (round 3): a 256 x 128 tile
// (waves 2 x 2, 128 x 64 each, 128 accumulators) -- 48 KiB per K-tile for twice the FLOPs of a 128 x 128 tile, i.e. 0.75 x
// the L2 -> LDS bytes per FLOP, which is what the 128 x 128 kernel is bound by on mid-size shapes
// (profiles/r02_lp128_mid_size_bound.md: 42-46 B/clk per CU of a 64 B/clk path with and without the matrix pipe running).
// Three-stage ring (144 KiB) + loader waves, one workgroup per CU; 16-bit operands only (fp8 fragments are twice as wide).
template <int MI> struct geom {
    static constexpr int BMK = 64 * MI;                        // tile rows
    static constexpr int A_BYTES = BMK * ROW_BYTES;            // A tile per stage
    static constexpr int STAGE = A_BYTES + TILE_BYTES;         // A + B per stage
};
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int DT> struct lp;
template <> struct lp<MI355_DTYPE_BF16> {
    typedef bf16x8 frag;
    static __device__ __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c)
    { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct lp<MI355_DTYPE_F16> {
    typedef f16x8 frag;
    static __device__ __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c)
    { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

// OCP FP8: one v_mfma_f32_32x32x64_f8f6f4 (64 cycles) per 32-byte fragment = two adjacent 16-byte chunks of the row; a
// K-tile is two such k-steps (see gemm_lp256w4.hip for the instruction; unscaled, any consistent k assignment is valid).
template <> struct lp<MI355_DTYPE_F8E4M3> {
    typedef i32x8 frag;
    static __device__ __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c)
    { return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0); }
};
template <> struct lp<MI355_DTYPE_F8E5M2> {
    typedef i32x8 frag;
    static __device__ __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c)
    { return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 1, 1, 0, 0, 0, 0); }
};

__device__ __forceinline__ void glds16(const void *gsrc, void *lds_dst)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)lds_dst, 16, 0, 0);
}

// NS = LDS stages.  1 (32-36 KiB, four workgroups per CU): launches whose K is a single K-tile -- output-bound, nothing to
// pipeline inside a workgroup, so what hides one workgroup's operand fetch and MFMAs is the C stores of the three others.
// 2 (64 KiB, two workgroups per CU): the co-resident workgroup covers the wait for the next K-tile.
// 4 (128 KiB, one workgroup per CU): taken when the launch has at most one workgroup per CU anyway (mid-size shapes:
// <= 256 tiles) -- then nothing else hides the DMA latency, and the K-tiles are fetched three ahead instead of one.
// LDS-DMA as `global_load_lds_dwordx4 v_off, s[base:base+1]`: wave-uniform 64-bit base + constant 32-bit lane offset, LDS
// destination through M0 (set in the same statement); see gemm_lp256w4.hip.  Not counted by the compiler: every wait on
// these loads in this file is explicit.
#ifndef LDS_DMA_POLICY
#define LDS_DMA_POLICY 0   // dev: cache-policy modifiers of the LDS-DMA loads: 1 sc0, 2 sc1, 3 sc0 sc1, 4 nt (measured: profiles/r03_lds_dma_cache_policy.md)
#endif
#if LDS_DMA_POLICY == 1
#define LDS_DMA_MOD " sc0"
#elif LDS_DMA_POLICY == 2
#define LDS_DMA_MOD " sc1"
#elif LDS_DMA_POLICY == 3
#define LDS_DMA_MOD " sc0 sc1"
#elif LDS_DMA_POLICY == 4
#define LDS_DMA_MOD " nt"
#else
#define LDS_DMA_MOD ""
#endif
template <bool NT = false>
__device__ __forceinline__ void glds16_s(const void *ubase_in, uint32_t voff, uint32_t lds_in)
{
    // wave-uniform by construction; said again here because hipcc's divergence analysis loses it behind role branches
    // (the "s" constraint then gets a VGPR pair and the assembler rejects the instruction).  Free when already scalar.
    const uint64_t u = reinterpret_cast<uint64_t>(ubase_in);
    const uint64_t us = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32)) << 32) |
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u);
    const void *ubase = reinterpret_cast<const void *>(us);
    const uint32_t lds_byte_addr = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_in);
    // s_nop 4: the base may have just been written by v_readfirstlane, and on gfx9 a VMEM instruction must not read an SGPR
    // within 5 wait states of a VALU write to it.  The compiler pads its own code for that hazard but cannot see into
    // inline asm (tools/hazard_scan.py, run by tests/test_abi_cpu.py: 2-3 wait states here before the padding; a prefetch
    // experiment with none faulted at once).
    // NT: an operand that no other tile shares (one tile row / column) and that is larger than the Infinity Cache could keep is
    // streamed non-temporally (gemm_args::nt_mask, set by the launcher; profiles/r03_lds_dma_cache_policy.md)
    if constexpr (NT)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(voff), "s"(ubase), "s"(lds_byte_addr) : "memory");
    else
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" LDS_DMA_MOD ::"v"(voff), "s"(ubase), "s"(lds_byte_addr) : "memory");
}
__device__ __forceinline__ uint32_t lds_addr_of(const void *p)
{
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char *)p;
}

// SPEC (with the 4-stage ring only): eight waves, four that multiply and four that do nothing but issue the LDS-DMA pieces
// and wait for them.  One `global_load_lds_dwordx4` costs the wave that issues it 60-185 cycles of its instruction stream
// (MI355X_MICROARCH.md, "LDS-DMA piece issue cost"); at eight pieces per wave and K-tile that is more than the 512 cycles
// the K-tile's 16 MFMAs take, so a wave that does both starves its matrix pipe.  The loader waves own `vmcnt`; the
// multiplying waves see the ring only through the one s_barrier per K-tile.
template <int DT, int DT_C, int NS = 2, bool SPEC = false, bool BNN = false, int MI = 2, bool ATN = false>
__global__ void __launch_bounds__(SPEC ? 512 : 256,
                                  // waves per SIMD the register budget must allow (fp8 fragments are twice as wide: three, not four)
                                  NS == 1 ? ((DT == MI355_DTYPE_F8E4M3 || DT == MI355_DTYPE_F8E5M2) ? 3 : 4) : NS == 2 ? (SPEC ? 4 : 2) : SPEC ? 2 : 1)
gemm_lp128_kernel(gemm_args g)
{
    static_assert(!SPEC || NS == 4 || NS == 3 || NS == 2, "loader waves are written for the 2-stage ring and the deep rings");
    static_assert(MI == 1 || MI == 2 || (MI == 4 && NS == 3 && DT != MI355_DTYPE_F8E4M3 && DT != MI355_DTYPE_F8E5M2), "256 x 128 tile: three-stage ring, 16-bit operands");
    constexpr int BMK = geom<MI>::BMK, A_BYTES = geom<MI>::A_BYTES, STG = geom<MI>::STAGE;
    constexpr int PIECES = 2 * MI + 4;           // LDS-DMA instructions per wave and K-tile (A: BMK / 32, B: 4)
    static_assert(!BNN || DT == MI355_DTYPE_BF16 || DT == MI355_DTYPE_F16, "row-major B: 16-bit operands");
    static_assert(!ATN || (BNN && MI == 2), "A stored [K][M]: together with a row-major B, 128 x 128 tile");
    // [stage][operand][16 KiB]; one array only (a second __shared__ object de-pipelines LDS-DMA
    // loops: guide section 5, ".s-level traps" (a))
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    typedef typename lp<DT>::frag frag;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = SPEC && wave_all >= 4;                 // waves 4..7: DMA issue only
    const int wave = SPEC ? (wave_all & 3) : wave_all;                             // position in the DMA map / in the 2 x 2 wave grid
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5, l31 = lane & 31;

    uint32_t tm, tn, batch_u;
    batched_tile_coords(g.tiles_m, g.tiles_n, g.group_m, tm, tn, batch_u);
    const int64_t m0 = (int64_t)tm * BMK, n0 = (int64_t)tn * BN;
    const int64_t batch = batch_u;
    constexpr bool F8 = DT == MI355_DTYPE_F8E4M3 || DT == MI355_DTYPE_F8E5M2;
    constexpr int ESZ = F8 ? 1 : 2;
    constexpr int BK = ROW_BYTES / ESZ;                  // k-values per K-tile: 128 (fp8) / 64 (16-bit)
    constexpr int NSTEP = F8 ? 2 : 4;                    // k-steps per K-tile (fp8: 64 k-values per MFMA)
    const char *__restrict__ A = static_cast<const char *>(g.a) + batch * g.stride_a * ESZ;
    const char *__restrict__ B = static_cast<const char *>(g.b) + batch * g.stride_b * ESZ;

    // ---- DMA map: wave w, instruction j fills rows (j*4+w)*8 .. +7 of the tile ----------------
    const char *ubase_a = ATN ? A + m0 * ESZ : A + m0 * g.lda * ESZ;              // uniform: first row of the tile (first column, A stored [K][M])
    const char *ubase_b = BNN ? B + n0 * ESZ : B + n0 * g.ldb * ESZ;              // ... first column, for row-major B
    uint32_t va[2 * MI], vb[4];                             // per-lane byte offsets from those (rows clamped at the edges)
#pragma unroll
    for (int j = 0; j < 2 * MI; ++j) {
        const int r = (j * 4 + wave) * 8 + (lane >> 3);    // tile row this lane fills
        const int q = (lane & 7) ^ ((r >> 1) & 7);          // logical chunk fetched into physical chunk lane&7
        if constexpr (ATN) {
            // A stored [K][M] (the lhs of a weight-gradient product, lhs^T . grad): the A tile is 64 k-rows x 128 m, the mirror
            // image of the row-major B tile -- same blocks of [4 k][32 m], same pieces (see vb below)
            const int64_t col = (lane >> 4) * 32 + (lane & 3) * 8;
            va[j] = (uint32_t)(((j * 4 + wave) * 4 + ((lane & 15) >> 2)) * g.lda * ESZ + min(col, g.m - 8 - m0) * ESZ);
        } else
            va[j] = (uint32_t)(min((int64_t)r, g.m - 1 - m0) * g.lda * ESZ + q * 16);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = (j * 4 + wave) * 8 + (lane >> 3);
        const int q = (lane & 7) ^ ((r >> 1) & 7);
        if constexpr (BNN) {
            // piece p = j*4 + wave is block row a = p (k-rows 4p .. 4p+3): lane -> block lane/16, row (lane%16)/4 of it,
            // 16-byte chunk lane%4 = columns 32 (lane/16) + 8 (lane%4) .. +7; columns past N re-read the last valid 16 bytes
            const int64_t col = (lane >> 4) * 32 + (lane & 3) * 8;
            vb[j] = (uint32_t)(((j * 4 + wave) * 4 + ((lane & 15) >> 2)) * g.ldb * ESZ + min(col, g.n - 8 - n0) * ESZ);
        } else
            vb[j] = (uint32_t)(min((int64_t)r, g.n - 1 - n0) * g.ldb * ESZ + q * 16);
    }
    // row-major B: this lane's part of a transposing read -- block wn*2 + j of a block row, row (lane%16)/4, 16-lane group, 8 B per lane
    const int rbn = BNN ? wn * 2 * 256 + ((lane & 15) >> 2) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8 : 0;
    const int ran = ATN ? wm * 2 * 256 + ((lane & 15) >> 2) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8 : 0;

    // ---- fragment read offsets (bytes inside one operand tile) --------------------------------
    int ra[MI], rb[2], fa[MI], fb[2];
#pragma unroll
    for (int t = 0; t < MI; ++t) {
        const int rowa = wm * 32 * MI + t * 32 + l31;
        ra[t] = rowa * ROW_BYTES; fa[t] = (rowa >> 1) & 7;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int rowb = wn * 64 + t * 32 + l31;
        rb[t] = rowb * ROW_BYTES; fb[t] = (rowb >> 1) & 7;
    }

    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // split-K: K slice z of g.split_k covers K-tiles [kt0, kt0 + nk); its partial goes to slab z
    const int nk_total = (int)(g.k / BK);
    const int z = (g.split_k > 1) ? (int)blockIdx.z : 0;
    const int per = (nk_total + (int)g.split_k - 1) / (int)max(g.split_k, 1u);
    const int kt0 = z * per;
    const int nk = max(0, min(per, nk_total - kt0));

#ifndef LP128_KSTAG
#define LP128_KSTAG 0   // dev: K-tile order rotated per workgroup (1: by tm + tn, 2: by (tm + tn) & 3, 3: by workgroup id)
#endif
#ifndef LP128_ABL
#define LP128_ABL 0     // dev, timing only: 1 = no DMA after the prologue, 2 = no fragment reads after the first, 4 = no MFMA,
                        // 8 = no C stores (staging kept), 16 = no epilogue at all
#endif
#ifndef LP128_NT
#define LP128_NT 1      // non-temporal C stores: 1 = every form (round 3); -1 = every form but the single-stage one (round 2); 0 never.
                        // Round 2 kept plain stores in the single-stage form on measurements with ONE output buffer (the 128 MiB of C stayed
                        // in the Infinity Cache: 27.1 us plain against 28.6 nt at 8192 x 8192 x 64).  With launches rotating through output
                        // buffers the cache only delays the write-back: 38.0 us plain against 31.8 nt, 8192 x 8192 x 128 47.6 / 39.5,
                        // 16384 x 4096 x 128 47.4 / 40.0 (interleaved, profiles/r03_output_bound_nt_stores.txt).
#endif
    const int kshift = nk <= 0 ? 0 : LP128_KSTAG == 1 ? (int)((tm + tn) % (uint32_t)nk) : LP128_KSTAG == 2 ? (int)((tm + tn) & 3u) % nk
                                   : LP128_KSTAG == 3 ? (int)(blockIdx.x % (uint32_t)nk) : 0;
#ifndef LP128_HYB
#define LP128_HYB 0     // dev, 256 x 128 tile: 1 = the multiplying waves issue the B pieces themselves (measured, slower: below)
#endif
    // HYB (dev switch, off): the four loader waves fetch the A tile only (8 pieces each per K-tile) and the four multiplying
    // waves the B tile (4 pieces each, issued behind the hand-over barrier).  Idea: twelve pieces per loader wave and K-tile
    // pace the 256 x 128 tile at ~125 cycles per piece, and a third wave per SIMD does not fit beside 196-VGPR multiplying
    // waves.  Measured, interleaved twice against the product (profiles/r03_tile_256x128.md): 4096 x 2048 x 4096 80.3 us against
    // 72.4, 2560^2 x 4096 64.8-72.8 / 58.4, 4096 x 2048 x 8192 143 / 129 -- an LDS-DMA piece costs the wave that issues it far
    // more than its slot between two MFMAs here (no scalar-base addressing behind the role branch), and the matrix pipe waits.
    constexpr bool HYB = LP128_HYB && MI == 4 && SPEC;
    constexpr int PIECES_L = HYB ? 2 * MI : PIECES;     // pieces per LOADER wave and K-tile
    auto stage = [&](int buf, int kt_rel, bool do_a = true, bool do_b = true) {
        if ((LP128_ABL & 1) && kt_rel >= NS) return;
        int kt = kt0 + kt_rel;
        if (LP128_KSTAG) { kt = kt_rel + kshift; kt = kt0 + (kt >= nk ? kt - nk : kt); }
        char *la = smem + buf * STG;
        char *lb = la + A_BYTES;
        const int64_t koff = (int64_t)kt * ROW_BYTES;
#pragma unroll
        for (int j = 0; j < (2 * MI > 4 ? 2 * MI : 4); ++j) {
            if (j < 2 * MI && do_a) {
                if (g.nt_mask & 1u) glds16_s<true>(ubase_a + (ATN ? koff * g.lda : koff), va[j], lds_addr_of(la + (j * 4 + wave) * 1024));    // (uniform branches)
                else glds16_s<false>(ubase_a + (ATN ? koff * g.lda : koff), va[j], lds_addr_of(la + (j * 4 + wave) * 1024));
            }
            if (j < 4 && do_b) {                                               // (row-major B: a K-tile is 64 rows of ldb elements)
                if (g.nt_mask & 2u) glds16_s<true>(ubase_b + (BNN ? koff * g.ldb : koff), vb[j & 3], lds_addr_of(lb + (j * 4 + wave) * 1024));
                else glds16_s<false>(ubase_b + (BNN ? koff * g.ldb : koff), vb[j & 3], lds_addr_of(lb + (j * 4 + wave) * 1024));
            }
        }
    };

#ifndef LP128_INSTREAM
#define LP128_INSTREAM 0   // dev, deep rings without loader waves: 1 = the multiplying waves issue the pieces of K-tile kt+AHEAD in four
                           // portions between the k-steps of the following K-tile (gemm_lp256w4.hip's in-stream issue) instead of
                           // back to back behind the hand-over barrier.  Round 4, interleaved three times on cold operands against the
                           // loader-wave forms (profiles/r04_lp128_in_stream_dma.txt): the 256 x 128 tile LOSES 9-17 % (4096 x 2048 x 4096 80 ->
                           // 87 us, 2560^2 x 4096 64 -> 75, 4096 x 1536 x 8192 114 -> 132), the 128 x 128 tile ties (2048^3 24.6 / 24.1,
                           // 1024 x 4096 x 4096 45.9 / 45.2).  Off; the loader waves stay.
#endif
    // linear piece p of a K-tile: 0 .. 2 MI - 1 = A piece p, then the four B pieces
    auto stage_pieces = [&](int buf, int kt_rel, int p0, int p1) {
        int kt = kt0 + kt_rel;
        char *la = smem + buf * STG;
        char *lb = la + A_BYTES;
        const int64_t koff = (int64_t)kt * ROW_BYTES;
#pragma unroll
        for (int p = 0; p < PIECES; ++p) {
            if (p < p0 || p >= p1) continue;
            if (p < 2 * MI) glds16_s<false>(ubase_a + (ATN ? koff * g.lda : koff), va[p < 2 * MI ? p : 0], lds_addr_of(la + (p * 4 + wave) * 1024));
            else glds16_s<false>(ubase_b + (BNN ? koff * g.ldb : koff), vb[(p - 2 * MI) & 3], lds_addr_of(lb + ((p - 2 * MI) * 4 + wave) * 1024));
        }
    };
    frag af[2][MI], bf[2][2];                    // [register buffer][tile]
    bool first_reads = true;
    auto reads = [&](auto buf, const char *la, const char *lb, int kk) {
        constexpr int B = decltype(buf)::value;
        if (LP128_ABL & 2) { if (!first_reads) return; if (B == 1) first_reads = false; }
        if constexpr (F8) {
            // k-step kk, lane-half h: logical chunks 4kk + 2h and 4kk + 2h + 1 (the second sits in physical chunk ^ 1)
            const int q = kk * 4 + 2 * h;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const char *p0 = la + ra[i] + ((q ^ fa[i]) << 4);
                const u32x4 lo = *reinterpret_cast<const u32x4 *>(p0);
                const u32x4 hi = *reinterpret_cast<const u32x4 *>(la + ra[i] + (((q ^ fa[i]) ^ 1) << 4));
                af[B][i] = (frag){(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const u32x4 lo = *reinterpret_cast<const u32x4 *>(lb + rb[j] + ((q ^ fb[j]) << 4));
                const u32x4 hi = *reinterpret_cast<const u32x4 *>(lb + rb[j] + (((q ^ fb[j]) ^ 1) << 4));
                bf[B][j] = (frag){(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
            }
        } else {
            const int q = kk * 2 + h;            // logical 16-byte chunk: 8 k-values
            typedef short s16x4 __attribute__((ext_vector_type(4)));
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            if constexpr (ATN) {
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const auto p = (__attribute__((address_space(3))) s16x4 *)(la + ran + (kk * 4 + 2 * h) * 1024 + i * 256);
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p + 128);
                    af[B][i] = __builtin_bit_cast(frag, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                }
            } else {
#pragma unroll
                for (int i = 0; i < MI; ++i) af[B][i] = *reinterpret_cast<const frag *>(la + ra[i] + ((q ^ fa[i]) << 4));
            }
            if constexpr (BNN) {
                // k-step kk, lane-half h: k 0..3 of its eight from block row a = 4kk + 2h, k 4..7 from a + 1 (4 blocks = 1 KiB on)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const auto p = (__attribute__((address_space(3))) s16x4 *)(lb + rbn + (kk * 4 + 2 * h) * 1024 + j * 256);
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p + 128);
                    bf[B][j] = __builtin_bit_cast(frag, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                }
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j) bf[B][j] = *reinterpret_cast<const frag *>(lb + rb[j] + ((q ^ fb[j]) << 4));
            }
        }
    };
    auto mfmas = [&](auto buf) {
        constexpr int B = decltype(buf)::value;
        if (LP128_ABL & 4) return;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < MI; ++i) acc[i][j] = lp<DT>::mfma(bf[B][j], af[B][i], acc[i][j]);
    };
    typedef std::integral_constant<int, 0> B0;
    typedef std::integral_constant<int, 1> B1;

    if constexpr (NS == 1) {
        // one stage: fetch, wait, multiply, hand the buffer back.  Nothing overlaps inside the workgroup; the three
        // co-resident workgroups do the overlapping (launched for K of at most SK1_MAX_TILES K-tiles).
        for (int kt = 0; kt < nk; ++kt) {
            if (kt > 0) __syncthreads();                     // every wave is done reading the previous K-tile
            stage(0, kt);
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            const char *la = smem, *lb = smem + A_BYTES;
            reads(B0{}, la, lb, 0);
            reads(B1{}, la, lb, 1); mfmas(B0{});
            if constexpr (NSTEP == 4) {
                reads(B0{}, la, lb, 2); mfmas(B1{});
                reads(B1{}, la, lb, 3); mfmas(B0{});
            }
            mfmas(B1{});
        }
    } else if constexpr (NS == 2 && SPEC) {
        // two stages, loader waves: the loaders fetch K-tile kt+1 while the others multiply K-tile kt; one raw barrier per
        // K-tile hands the buffers over (the loaders wait for their DMA before it, the multiplying waves have consumed their
        // fragment reads before it)
        if (loader) {
            if (nk > 0) stage(0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            for (int kt = 0; kt < nk; ++kt) {
                if (kt + 1 < nk) stage((kt & 1) ^ 1, kt + 1);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
            }
            return;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);                     // see the 4-stage form: keeps the loop's LDS waits counted
        __builtin_amdgcn_s_barrier();
        for (int kt = 0; kt < nk; ++kt) {
            const char *la = smem + (kt & 1) * STG;
            const char *lb = la + A_BYTES;
            reads(B0{}, la, lb, 0);
            reads(B1{}, la, lb, 1); mfmas(B0{});
            if constexpr (NSTEP == 4) {
                reads(B0{}, la, lb, 2); mfmas(B1{});
                reads(B1{}, la, lb, 3); mfmas(B0{});
            }
            mfmas(B1{});
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        }
    } else if constexpr (NS == 2) {
        if (nk > 0) stage(0, 0);
        __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) lgkmcnt(0) expcnt(0)
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
            const char *la = smem + cur * STG;
            const char *lb = la + A_BYTES;
            // fragments double-buffered in registers: the reads of k-step kk+1 go out before the MFMAs of k-step kk
            // (+2...5 % even with the co-resident workgroup filling gaps)
            reads(B0{}, la, lb, 0);
            reads(B1{}, la, lb, 1); mfmas(B0{});
            if constexpr (NSTEP == 4) {
                reads(B0{}, la, lb, 2); mfmas(B1{});
                reads(B1{}, la, lb, 3); mfmas(B0{});
            }
            mfmas(B1{});
            // the DMA for tile kt+1 must have landed and every wave must be done reading `cur`
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
        }
    } else {
        // Deep ring (NS stages, K-tiles fetched NS-1 ahead) with the fragments double-buffered in registers: the reads
        // of k-step kk+1 are issued before the MFMAs of k-step kk, and the hand-over to the next K-tile sits before the
        // LAST k-step, so that the next tile's first fragments are fetched under its four MFMAs (as gemm_lp256w4.hip).
        // vmcnt is counted: PIECES DMA instructions per wave and K-tile, loads complete in order.  K-tiles are issued three
        // ahead in both forms: K-tile kt+2 may still fly when K-tile kt+1 is needed.
        //   NS = 4: K-tile kt+3 is issued after the hand-over barrier of K-tile kt into the slot of K-tile kt-1 -- the
        //           multiplying waves may still have fragment reads of K-tile kt in the LDS queue at that barrier.
        //   NS = 3 (256 x 128 tile; 48 KiB per stage leave room for three): K-tile kt+3 goes into the slot of K-tile kt
        //           ITSELF, so the multiplying waves wait for their last reads of it (lgkmcnt(0)) before the barrier, as
        //           gemm_lp256w4.hip does.  (The first form of this kernel refilled the slot of K-tile kt-1 with K-tile kt+2:
        //           one K-tile in flight, fetch and wait serialised, ~2 000 cycles per K-tile of 1 024 MFMA cycles.)
        static_assert(NS == 4 || NS == 3, "the counted waits below are written for three or four stages");
#ifndef LP128_FULL4
#define LP128_FULL4 0   // dev: the four-stage ring refilled in place as well (four K-tiles ahead); interleaved twice over ten skinny / mid-size
                        // shapes on cold operands: no effect beyond run-to-run noise (+-5 %), as round 2's five-stage ring
#endif
        constexpr bool FULL = NS == 3 || LP128_FULL4;  // every slot of the ring is refilled as soon as its K-tile is consumed
        constexpr int AHEAD = FULL ? NS : NS - 1;      // K-tiles issued ahead of the one being multiplied
        // at most `n` whole K-tiles of this wave's pieces may still be in flight
        auto wait_tiles = [](int n, auto pieces) {
            constexpr int P = decltype(pieces)::value;
            if (n >= 3) wait_vm<3 * P>(); else if (n == 2) wait_vm<2 * P>(); else if (n == 1) wait_vm<P>(); else wait_vm<0>();
        };
        const bool issues = !SPEC || loader, multiplies = !SPEC || !loader;
        if (issues) {
#pragma unroll
            for (int p = 0; p < AHEAD; ++p)
                if (p < nk) stage(p % NS, p, true, !HYB);
            wait_tiles(min(nk, AHEAD) - 1, std::integral_constant<int, PIECES_L>{});     // K-tile 0 landed; the other prologue tiles may fly
        } else if constexpr (HYB) {                                                   // the multiplying waves' share: the B tiles
#pragma unroll
            for (int p = 0; p < AHEAD; ++p)
                if (p < nk) stage(p % NS, p, false, true);
            wait_tiles(min(nk, AHEAD) - 1, std::integral_constant<int, 4>{});
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (SPEC && loader) {
            // the loader's whole K loop: wait for K-tile kt+1, meet the multiplying waves at their barrier inside K-tile kt
            // (everybody is then past K-tile kt-1 -- FULL: and done reading K-tile kt), refill the freed slot with K-tile kt+3
            for (int kt = 0; kt + 1 < nk; ++kt) {
                wait_tiles(min(AHEAD - 2, nk - kt - 2), std::integral_constant<int, PIECES_L>{});   // K-tile kt+1 landed; the younger ones may fly
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if (kt + AHEAD < nk) stage((kt + AHEAD) % NS, kt + AHEAD, true, !HYB);
            }
            return;                                         // a finished wave no longer counts at the workgroup's barriers
        }
        if (multiplies) {
            // Scalar loads share lgkmcnt with the LDS and return out of order.  With one of them in flight at the loop header
            // -- as far as the compiler's wait-count bookkeeping knows: hence the builtin, which it reads, not inline asm --
            // every LDS wait of the loop degrades to lgkmcnt(0), fresh reads included.
            __builtin_amdgcn_s_waitcnt(0xC07F);                 // lgkmcnt(0), vmcnt / expcnt untouched
            __builtin_amdgcn_sched_barrier(0);
            if (nk > 0) reads(B0{}, smem, smem + A_BYTES, 0);
            constexpr bool INSTREAM = LP128_INSTREAM && !SPEC && NSTEP == 4;
            constexpr int Q = (PIECES + 3) / 4;          // pieces per portion
            int pend = -1;                               // K-tile whose pieces 1 .. 3 x Q are still to be issued (in-stream form)
            for (int kt = 0; kt < nk; ++kt) {
                const char *la = smem + (kt % NS) * STG;
                const char *lb = la + A_BYTES;
                reads(B1{}, la, lb, 1); mfmas(B0{});
                if constexpr (INSTREAM) if (pend >= 0) stage_pieces(pend % NS, pend, Q, 2 * Q);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (NSTEP == 4) {
                    reads(B0{}, la, lb, 2); mfmas(B1{});
                    if constexpr (INSTREAM) if (pend >= 0) stage_pieces(pend % NS, pend, 2 * Q, 3 * Q);
                    __builtin_amdgcn_sched_barrier(0);
                    reads(B1{}, la, lb, 3); mfmas(B0{});
                    if constexpr (INSTREAM) if (pend >= 0) { stage_pieces(pend % NS, pend, 3 * Q, PIECES); pend = -1; }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (kt + 1 < nk) {
                    if constexpr (!SPEC) {
                        // K-tile kt+1 landed: only tile kt+2 (issued in the previous iteration or the prologue) may still fly
                        wait_tiles(min(AHEAD - 2, nk - kt - 2), std::integral_constant<int, PIECES>{});
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if constexpr (HYB) {            // my B pieces of K-tile kt+1 have landed; those of K-tile kt+2 may fly
                        wait_tiles(min(AHEAD - 2, nk - kt - 2), std::integral_constant<int, 4>{});
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if constexpr (FULL) {           // my reads of K-tile kt are complete: its slot is refilled right behind the barrier
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    __builtin_amdgcn_s_barrier();   // raw: __syncthreads() carries a release fence = vmcnt(0), which would drain the ring
                    __builtin_amdgcn_sched_barrier(0);
                    // everybody is past K-tile kt-1: its buffer takes K-tile kt+3
                    if constexpr (!SPEC && !INSTREAM)
                        if (kt + AHEAD < nk) stage((kt + AHEAD) % NS, kt + AHEAD);
                    if constexpr (INSTREAM)           // the first portion here, the other three between the next K-tile's k-steps
                        if (kt + AHEAD < nk) { pend = kt + AHEAD; stage_pieces(pend % NS, pend, 0, Q); }
                    if constexpr (HYB)
                        if (kt + AHEAD < nk) stage((kt + AHEAD) % NS, kt + AHEAD, false, true);
                    const char *na = smem + ((kt + 1) % NS) * STG;
                    reads(B0{}, na, na + A_BYTES, 0);
                }
                mfmas(B1{});
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // ---- epilogue ---------------------------------------------------------------------------
    if (LP128_ABL & 16) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int i = 0; i < MI; ++i) t += acc[i][0][r] + acc[i][1][r];
        if (t == 12345.678f) static_cast<float *>(g.c)[0] = t;
        return;
    }
    char *__restrict__ C = static_cast<char *>(g.c);
    constexpr int CSZ = (DT_C == MI355_DTYPE_F32) ? 4 : 2;
    const int64_t cbase = batch * g.stride_c + (int64_t)z * g.split_c_stride;
    // Whole-line stores.  Straight from the accumulators a store instruction would put 8 (16-bit) or 16 (f32) bytes on
    // each of 32 different rows, so every 128-byte line of C would be assembled from 8-16 partial writes -- measured on the
    // 256x256 kernel and again here, partial-line stores cost tens of percent (8192 x 8192 x 64 bf16: 49.6 us before).  Each
    // wave instead transposes its 64 x 64 block through LDS, 32 rows at a time, and writes rows: 16 bytes per lane,
    // 128 (16-bit) / 256 (f32) contiguous bytes per row.  Needs 16-byte aligned rows; anything else takes the element path.
    const bool rows16 = (((g.ldc * CSZ) & 15) == 0) && (((reinterpret_cast<uintptr_t>(C) + (uint64_t)cbase * CSZ) & 15u) == 0);
    if (rows16) {
        constexpr int RS = 64 * CSZ + 16;                  // staged row pitch: +16 B keeps b128 aligned, writes <= 2-way conflicted
        constexpr int STAGE = (32 * RS + 1023) & ~1023;    // per-wave scratch
        constexpr int LPR = 64 * CSZ / 16;                 // lanes per output row: 8 / 16
        constexpr int RPI = 64 / LPR;                      // rows per store instruction: 8 / 4
        constexpr int EPP = 16 / CSZ;                      // elements per 16-byte piece
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();                                   // every wave is done with the operand tiles: LDS is free
        char *stage = smem + wave * STAGE;
        char *wr = stage + l31 * RS + 4 * h * CSZ;
        const char *rd = stage + (lane / LPR) * RS + (lane % LPR) * 16;
        const int64_t row0 = m0 + wm * 32 * MI + lane / LPR;              // + i * 32 + it * RPI
        const int64_t col0 = n0 + wn * 64 + (lane % LPR) * EPP;
        char *crow = C + (cbase + row0 * g.ldc + col0) * CSZ;
        const int64_t cstep = (int64_t)RPI * g.ldc * CSZ;
        const int ncols = (int)max((int64_t)0, min((int64_t)EPP, g.n - col0));   // valid elements of my piece
        const bool interior = (m0 + BMK <= g.m) && (n0 + BN <= g.n);      // workgroup-uniform fast path
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    char *d = wr + (j * 32 + 8 * q) * CSZ;
                    if constexpr (DT_C == MI355_DTYPE_F32) {
                        f32x4 v = {acc[i][j][4 * q + 0], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                        *reinterpret_cast<f32x4 *>(d) = v;
                    } else {
                        u32x2 v = {f32x2_to_lp<DT_C>(acc[i][j][4 * q + 0], acc[i][j][4 * q + 1]),
                                   f32x2_to_lp<DT_C>(acc[i][j][4 * q + 2], acc[i][j][4 * q + 3])};
                        *reinterpret_cast<u32x2 *>(d) = v;
                    }
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // same-wave hand-over: the DS ops of one wave execute in order
            char *cdst = crow + (int64_t)i * 32 * g.ldc * CSZ;
#pragma unroll
            for (int it = 0; it < 32 / RPI; ++it) {
                const u32x4 v = *reinterpret_cast<const u32x4 *>(rd + it * RPI * RS);
                if (!interior) {
                    if (row0 + i * 32 + it * RPI >= g.m || ncols <= 0) continue;
                    if (ncols < EPP) {
#pragma unroll
                        for (int e = 0; e < EPP; ++e) {                    // static indices only
                            if (e >= ncols) break;
                            if constexpr (CSZ == 4) reinterpret_cast<uint32_t *>(cdst + it * cstep)[e] = v[e];
                            else reinterpret_cast<uint16_t *>(cdst + it * cstep)[e] = (uint16_t)(v[e >> 1] >> ((e & 1) * 16));
                        }
                        continue;
                    }
                }
                if ((LP128_ABL & 8) && v[0] != 0x12345678u) continue;
                if (LP128_NT == 1 || (LP128_NT < 0 && NS != 1)) __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(cdst + it * cstep));
                else *reinterpret_cast<u32x4 *>(cdst + it * cstep) = v;
            }
            __builtin_amdgcn_sched_barrier(0);             // keep the accumulator reads of block i+1 below this point
        }
        return;
    }
    const bool vec_ok = false;   // rows are not 16-byte aligned here
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int64_t m = m0 + wm * 32 * MI + i * 32 + l31;
        if (m >= g.m) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t n = n0 + wn * 64 + j * 32 + 8 * q + 4 * h;
                const int64_t idx = cbase + m * g.ldc + n;
                if (DT_C == MI355_DTYPE_F32) {
                    float *dst = reinterpret_cast<float *>(C) + idx;
                    if (vec_ok && n + 3 < g.n) {
                        f32x4 v = {acc[i][j][4 * q + 0], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                        *reinterpret_cast<f32x4 *>(dst) = v;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (n + r < g.n) dst[r] = acc[i][j][4 * q + r];
                    }
                } else {
                    uint16_t *dst = reinterpret_cast<uint16_t *>(C) + idx;
                    uint16_t o[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = f32_to_lp<DT_C>(acc[i][j][4 * q + r]);
                    if (vec_ok && n + 3 < g.n) {
                        u32x2 v = {(uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16)};
                        *reinterpret_cast<u32x2 *>(dst) = v;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (n + r < g.n) dst[r] = o[r];
                    }
                }
            }
        }
    }
}

#ifndef LP128_SPEC2
#define LP128_SPEC2 1  // dev: 0 = the two-stage form without loader waves
#endif
#ifndef LP128_SPEC
#define LP128_SPEC 1   // dev: 0 = the 4-stage ring without loader waves
#endif
template <int DT, int DT_C, int NS, bool SPEC = false, bool BNN = false, int MI = 2, bool ATN = false>
void launch_ns(mi355_ctx *ctx, hipStream_t s, const gemm_args &g, uint32_t batch)
{
    // operand stages, or the four per-wave epilogue scratch areas when those are larger (f32 C with one stage: 36 KiB)
    constexpr int CSZ_ = DT_C == MI355_DTYPE_F32 ? 4 : 2;
    constexpr int EPI = 4 * ((32 * (64 * CSZ_ + 16) + 1023) & ~1023);
    constexpr int LDS = NS * geom<MI>::STAGE > EPI ? NS * geom<MI>::STAGE : EPI;
    lds_opt_in(ctx, reinterpret_cast<const void *>(gemm_lp128_kernel<DT, DT_C, NS, SPEC, BNN, MI, ATN>), LDS);
    hipLaunchKernelGGL((gemm_lp128_kernel<DT, DT_C, NS, SPEC, BNN, MI, ATN>), dim3(g.tiles_m * g.tiles_n, batch, g.split_k > 1 ? g.split_k : 1),
                       dim3(SPEC ? 512 : 256), LDS, s, g);
}

#ifndef SK1_MAX_TILES
#define SK1_MAX_TILES 4   // K-tiles up to which the single-stage, four-workgroups-per-CU form is launched
#endif
// At most 64 rows of A (16-bit operands): a 64 x 128 tile (MI = 1: waves 2 x 2, 32 x 64 each) -- the 128-row tile fetched its
// upper 64 rows as clamped duplicates from L2, 16 KiB of LDS-DMA intake per K-tile that multiply nothing.  24 KiB per K-tile
// instead of 32: interleaved twice on cold operands (profiles/r03_small_m_tile.txt), row-major weights 1 x 8192 x 8192 35.2 ->
// 32.3 us, 16 x 28672 x 8192 85.3 -> 76.3, 64 x 28672 x 8192 88.8 -> 82.6, 8 x 57344 x 4096 84.3 -> 75.6, 48 x 14336 x 4096 35.3 -> 31.0;
// [N][K] weights 64 x 28672 x 8192 93.5 -> 82.4-92.0, 64 x 14336 x 4096 36.8 -> 33.5-34.4 (every shape -5 ... -12 %).
#ifndef LP128_SMALL_M
#define LP128_SMALL_M 1   // dev: 0 = the 128 x 128 tile for every M
#endif
template <int DT, int DT_C, bool BNN = false, bool ATN = false>
void launch(mi355_ctx *ctx, hipStream_t s, const gemm_args &g, uint32_t batch)
{
    if constexpr (LP128_SMALL_M && !ATN && (DT == MI355_DTYPE_BF16 || DT == MI355_DTYPE_F16)) {
        if (g.m <= 64) {
            const uint64_t wgs1 = (uint64_t)g.tiles_m * g.tiles_n * batch * (g.split_k > 1 ? g.split_k : 1);
            if (wgs1 <= (uint64_t)ctx->props.num_streaming_multiprocessors) launch_ns<DT, DT_C, 4, LP128_SPEC != 0, BNN, 1, false>(ctx, s, g, batch);
            else launch_ns<DT, DT_C, 2, LP128_SPEC2 != 0, BNN, 1, false>(ctx, s, g, batch);
            return;
        }
    }
    // one workgroup per CU at most: the deep (4-stage) pipeline; otherwise two co-resident 2-stage workgroups per CU
    const uint64_t wgs = (uint64_t)g.tiles_m * g.tiles_n * batch * (g.split_k > 1 ? g.split_k : 1);
    constexpr int BK_ = ROW_BYTES / ((DT == MI355_DTYPE_F8E4M3 || DT == MI355_DTYPE_F8E5M2) ? 1 : 2);
    if (wgs <= (uint64_t)ctx->props.num_streaming_multiprocessors) launch_ns<DT, DT_C, 4, LP128_SPEC != 0, BNN, 2, ATN>(ctx, s, g, batch);
    else if (g.k <= SK1_MAX_TILES * BK_ && g.split_k <= 1) launch_ns<DT, DT_C, 1, false, BNN, 2, ATN>(ctx, s, g, batch);   // four workgroups per CU
    else launch_ns<DT, DT_C, 2, LP128_SPEC2 != 0, BNN, 2, ATN>(ctx, s, g, batch);
}

}  // namespace

namespace mi355 {

bool gemm_lp128_supports(const mi355_gemm_desc &d, const void *a, const void *b, const void *c)
{
    (void)c;
    const bool f8 = is_fp8(d.dtype_ab);
    if (d.dtype_ab != MI355_DTYPE_BF16 && d.dtype_ab != MI355_DTYPE_F16 && !f8) return false;
    if (f8) {
        if (d.dtype_c != MI355_DTYPE_F32 && d.dtype_c != MI355_DTYPE_BF16 && d.dtype_c != MI355_DTYPE_F16) return false;
    } else if (d.dtype_c != MI355_DTYPE_F32 && d.dtype_c != d.dtype_ab) return false;
    if (d.trans_a && (f8 || d.trans_b || d.m < 8 || (d.m & 7))) return false; // A stored [K][M]: 16-bit, with a row-major B, fetched 8 rows of C at a time
    if (!d.trans_b && (f8 || d.n < 8 || (d.n & 7))) return false;            // row-major B: 16-bit operands, fetched 8 columns (16 bytes) at a time
    const int64_t esz = f8 ? 1 : 2, BK = ROW_BYTES / esz, amask = 16 / esz - 1;
    if (d.k < BK || d.k % BK != 0) return false;
    if (d.m < 1 || d.n < 1) return false;
    if ((d.lda & amask) || (d.ldb & amask) || (d.stride_a & amask) || (d.stride_b & amask)) return false;
    if ((reinterpret_cast<uintptr_t>(a) & 15u) || (reinterpret_cast<uintptr_t>(b) & 15u)) return false;
    if (d.batch > 65535) return false;
    const int64_t tiles = ((d.m + BM - 1) / BM) * ((d.n + BN - 1) / BN);
    if (tiles * std::max<int64_t>(d.batch, 1) > 0x7FFFFFFF) return false;   // 32-bit (batch, tile) sequence for the XCD remap
    if ((int64_t)BM * std::max(d.lda, d.ldb) * esz >= (1ll << 32)) return false;   // per-lane DMA offsets are 32-bit
    return true;
}

// ---- how many K slices the launcher cuts a descriptor into (1 = none): a pure function of the descriptor and the number of CUs, so that the
// decisions below -- five of them were wrong at some point of round 3 -- are testable without a device (mi355_gemm_split_plan) ----------
int64_t lp128_split_count(const mi355_gemm_desc &d, int64_t cus)
{
    // Split-K for shapes whose tile count cannot fill the chip (skinny M or N, GEMV-like): K is cut into slices,
    // every slice writes an f32 partial slab, a small kernel folds the slabs in slice order (deterministic) and
    // converts.  The slab traffic (2 x splits x M x N x 4 B) must stay small against the operand stream.
    const int64_t batch = d.batch;
    const int64_t tiles = ((d.m + BM - 1) / BM) * ((d.n + BN - 1) / BN) * batch;
    const bool f8 = is_fp8(d.dtype_ab);
    const int64_t esz = f8 ? 1 : 2;
    const int64_t nk = d.k / (ROW_BYTES / esz);
#ifndef LP128_WANT_MULT
#define LP128_WANT_MULT 2   // workgroups the split aims at, in units of CUs
#endif
    // Slices so that tiles x slices fills the chip ONCE (one workgroup per CU, the deep ring) up to 96 tiles, twice (two co-resident
    // workgroups per CU) from there to 128 -- until late round 3 always twice.  What a split buys is idle CUs put to work; past one
    // workgroup per CU it only shortens the slices (pipeline fill per slice, more slab traffic) while the CU's LDS-DMA intake stays
    // what it is.  Interleaved twice on cold operands, both rhs layouts (profiles/r03_want_mult_nt_nn.txt, r03_want_mult_few_rows.txt):
    // 256 x 2048 x 8192 31-35 -> 23-24 us, 1024 x 512 x 8192 26.0 -> 22.3, 1024 x 1536 x 4096 31.5 -> 28.5, 128 x 8192 x 8192 47 -> 40-45,
    // row-major weights 64 x 8192 x 8192 40.0 -> 33.4, 48 x 14336 x 4096 31.5 -> 28.3, 1 x 8192 x 8192 33 -> 30; ties at 512^2 x 8192,
    // 1024^2 x 4096, 128 x 256 x 8192; 128 tiles keep the pair (2048 x 1024 x 4096 33 us against 39).
#ifndef LP128_ONCE_UPTO
#define LP128_ONCE_UPTO 96      // dev: 0 = round 3's earlier rule (always two workgroups per CU)
#endif
#ifndef LP128_NOSPLIT_FROM
#define LP128_NOSPLIT_FROM 160  // dev: 1 << 30 = the earlier rule (any tile count splits when K is long against it)
#endif
    const int64_t want = (tiles <= LP128_ONCE_UPTO ? 1 : LP128_WANT_MULT) * cus;
#ifndef LP128_SPLIT_RULE
#define LP128_SPLIT_RULE 1   // dev: 0 = round 1's rule (any launch of fewer than one tile per CU with 8+ K-tiles)
#endif
    // Splitting pays for its slabs (f32 partials written and read back, one more launch: 10-20 us on these shapes) only when
    // K is long against the number of tiles -- with T tiles, T CUs already stream 16 KiB per K-tile each -- and long in
    // absolute terms.  Measured pairs, bf16, no split / split (tools/dev/split_ab.py): 32 tiles x 32 K-tiles 14.1 / 17.8 us;
    // 48 x 48 21.4 / 20.6; 64 x 32 14.8 / 20.7, 64 x 64 26.3 / 25.3, 64 x 128 46.7 / 34.5; 96 x 64 25.9 / 34.8, 96 x 256
    // 161 / 131; 112 x 16 9.3 / 18.3, 112 x 64 26.3 / 38.9; 128 x 16 9.9 / 20.6, 128 x 128 50.5 / 46.5; 160 x 32 16.6 /
    // 34.5; 192 x 64 28 / 49; 224 x 128 (128 x 28672 x 8192) 104 / 142.  Rule: at least as many K-tiles as tiles, and 48 of
    // them unless the tiles are a handful.
    // Round 3, re-measured on cold operands with the slice count CAPPED by the slab-traffic bound (below) instead of the split
    // being rejected beyond it (profiles/r03_split_rule_cold.txt, rule against "always split", interleaved twice): with at most
    // 128 tiles splitting never loses and wins where the bound leaves two or more slices (32 tiles x 32 K-tiles 20.7 -> 15.3 us,
    // 64 x 32 20.6 -> 17.6, 72 x 64 36.5 -> 30.1, 96 x 64 37.5 -> 30.7, 128 x 64 37.4 -> 33.2); from 160 tiles up it loses unless K is
    // long against the tile count (192 x 64 38.6 against 50.4 split, 224 x 128 79 / 110).
    // (late round 3: from 160 tiles up never -- 1536 x 2048 x 16384, 192 tiles x 256 K-tiles, 176-183 us in two slices against
    //  150-155 unsplit; 768 x 3072 x 14336, 144 x 224, still wins split three ways: 122 against 136)
    const bool long_k = LP128_SPLIT_RULE == 0 || tiles <= 128 || (tiles < LP128_NOSPLIT_FROM && nk >= tiles && nk >= 48);
    if (!(tiles < std::max<int64_t>(want, 2 * cus) / 2 && nk >= 8 && long_k && batch <= 65535)) return 1;
    {
        const int64_t slab = d.batch * d.m * d.n;
        const int64_t operand_bytes = (d.m * d.k + d.n * d.k) * esz * d.batch;
        // The slab traffic (one f32 write + one read per slice and output) must stay below twice the operand stream.  Until
        // round 3 a split count beyond that bound was REJECTED (no split at all) instead of capped: 512 x 512 x 8192 wanted
        // 32 slices, was allowed 16 and ran on 16 workgroups -- 68.7 us against 20.7 with 16 slices; 1024^2 x 4096 37.7 -> 21.6,
        // 1024 x 512 x 8192 69 -> 25.5 (cold operands, interleaved; profiles/r03_split_k_cap.txt).
        const int64_t by_traffic = slab > 0 ? (2 * operand_bytes) / (slab * 8) : 1;
        // slices so that tiles x slices stays WITHIN the two-workgroups-per-CU residency (floor, not ceil: 144 tiles x 4 slices = 576
        // workgroups ran a partial second round -- 768 x 3072 x 14336 151 -> 118 us, 1536 x 2048 x 16384 205 -> 177 with 3 and 2 slices;
        // profiles/r03_split_count_floor.txt)
        int64_t splits = std::min<int64_t>({std::max<int64_t>(want / tiles, 1), nk / 4, 32, by_traffic});
        if (splits > 1) {
            const int64_t per = (nk + splits - 1) / splits;
            splits = (nk + per - 1) / per;                                  // no empty slices
        }
        return (splits > 1 && splits * slab * 8 <= 2 * operand_bytes) ? splits : 1;
    }
}

int32_t launch_gemm_lp128(mi355_ctx *ctx, hipStream_t s, const mi355_gemm_desc &d, const void *a, const void *b,
                          void *c)
{
    if (!gemm_lp128_supports(d, a, b, c))
        return fail(ctx, MI355_E_UNSUPPORTED, "lp128 GEMM: shape/layout not supported by this kernel");
    gemm_args g{};
    g.a = a; g.b = b; g.c = c;
    g.m = d.m; g.n = d.n; g.k = d.k;
    g.lda = d.lda; g.ldb = d.ldb; g.ldc = d.ldc;
    g.stride_a = d.stride_a; g.stride_b = d.stride_b; g.stride_c = d.stride_c;
    g.tiles_m = (uint32_t)((d.m + BM - 1) / BM);
    g.tiles_n = (uint32_t)((d.n + BN - 1) / BN);
    g.group_m = 8;
    g.split_k = 1;
    g.split_c_stride = 0;
    {
        // an operand whose tiles no other tile row / column shares, larger than the 256 MiB Infinity Cache could keep: read once
        const int64_t esz_ = is_fp8(d.dtype_ab) ? 1 : 2, nb = std::max<int64_t>(d.batch, 1);
#ifndef LP128_NO_NT        // dev: the A/B build without the hint
        if (g.tiles_n == 1 && d.m * d.k * esz_ * nb > (192ll << 20)) g.nt_mask |= 1u;
        if (g.tiles_m == 1 && d.n * d.k * esz_ * nb > (192ll << 20)) g.nt_mask |= 2u;
#endif
    }
    const uint32_t batch = (uint32_t)d.batch;
    // Split-K for shapes whose tile count cannot fill the chip: see lp128_split_count above
    const int64_t splits = lp128_split_count(d, (int64_t)ctx->props.num_streaming_multiprocessors);
    if (splits > 1) {
        const int64_t slab = d.batch * d.m * d.n;
        float *ws = nullptr;
        if (splitk_scratch(ctx, s, (size_t)(splits * slab) * sizeof(float), &ws) == MI355_OK) {
            gemm_args gs = g;
            gs.c = ws; gs.ldc = d.n; gs.stride_c = d.m * d.n;
            gs.split_k = (uint32_t)splits; gs.split_c_stride = slab;
            if (d.dtype_ab == MI355_DTYPE_BF16) { if (d.trans_a) launch<MI355_DTYPE_BF16, MI355_DTYPE_F32, true, true>(ctx, s, gs, batch); else if (d.trans_b) launch<MI355_DTYPE_BF16, MI355_DTYPE_F32>(ctx, s, gs, batch); else launch<MI355_DTYPE_BF16, MI355_DTYPE_F32, true>(ctx, s, gs, batch); }
            else if (d.dtype_ab == MI355_DTYPE_F16) { if (d.trans_a) launch<MI355_DTYPE_F16, MI355_DTYPE_F32, true, true>(ctx, s, gs, batch); else if (d.trans_b) launch<MI355_DTYPE_F16, MI355_DTYPE_F32>(ctx, s, gs, batch); else launch<MI355_DTYPE_F16, MI355_DTYPE_F32, true>(ctx, s, gs, batch); }
            else if (d.dtype_ab == MI355_DTYPE_F8E4M3) launch<MI355_DTYPE_F8E4M3, MI355_DTYPE_F32>(ctx, s, gs, batch);
            else launch<MI355_DTYPE_F8E5M2, MI355_DTYPE_F32>(ctx, s, gs, batch);
            check_launch(ctx, "mi355_gemm(lp128 split-K)");
            launch_splitk_fold(s, ws, (uint32_t)splits, slab, d.batch, d.m, d.n, c, d.dtype_c, d.ldc, d.stride_c);
            check_launch(ctx, "mi355_gemm(split-K fold)");
            return MI355_OK;
        }
    }
    if (d.dtype_ab == MI355_DTYPE_F8E4M3) {
        if (d.dtype_c == MI355_DTYPE_F32) launch<MI355_DTYPE_F8E4M3, MI355_DTYPE_F32>(ctx, s, g, batch);
        else if (d.dtype_c == MI355_DTYPE_BF16) launch<MI355_DTYPE_F8E4M3, MI355_DTYPE_BF16>(ctx, s, g, batch);
        else launch<MI355_DTYPE_F8E4M3, MI355_DTYPE_F16>(ctx, s, g, batch);
    } else if (d.dtype_ab == MI355_DTYPE_F8E5M2) {
        if (d.dtype_c == MI355_DTYPE_F32) launch<MI355_DTYPE_F8E5M2, MI355_DTYPE_F32>(ctx, s, g, batch);
        else if (d.dtype_c == MI355_DTYPE_BF16) launch<MI355_DTYPE_F8E5M2, MI355_DTYPE_BF16>(ctx, s, g, batch);
        else launch<MI355_DTYPE_F8E5M2, MI355_DTYPE_F16>(ctx, s, g, batch);
    } else if (d.dtype_ab == MI355_DTYPE_BF16) {
        if (d.trans_a) {
            if (d.dtype_c == MI355_DTYPE_F32) launch<MI355_DTYPE_BF16, MI355_DTYPE_F32, true, true>(ctx, s, g, batch);
            else launch<MI355_DTYPE_BF16, MI355_DTYPE_BF16, true, true>(ctx, s, g, batch);
        } else if (d.trans_b) {
            if (d.dtype_c == MI355_DTYPE_F32) launch<MI355_DTYPE_BF16, MI355_DTYPE_F32>(ctx, s, g, batch);
            else launch<MI355_DTYPE_BF16, MI355_DTYPE_BF16>(ctx, s, g, batch);
        } else {
            if (d.dtype_c == MI355_DTYPE_F32) launch<MI355_DTYPE_BF16, MI355_DTYPE_F32, true>(ctx, s, g, batch);
            else launch<MI355_DTYPE_BF16, MI355_DTYPE_BF16, true>(ctx, s, g, batch);
        }
    } else {
        if (d.trans_a) {
            if (d.dtype_c == MI355_DTYPE_F32) launch<MI355_DTYPE_F16, MI355_DTYPE_F32, true, true>(ctx, s, g, batch);
            else launch<MI355_DTYPE_F16, MI355_DTYPE_F16, true, true>(ctx, s, g, batch);
        } else if (d.trans_b) {
            if (d.dtype_c == MI355_DTYPE_F32) launch<MI355_DTYPE_F16, MI355_DTYPE_F32>(ctx, s, g, batch);
            else launch<MI355_DTYPE_F16, MI355_DTYPE_F16>(ctx, s, g, batch);
        } else {
            if (d.dtype_c == MI355_DTYPE_F32) launch<MI355_DTYPE_F16, MI355_DTYPE_F32, true>(ctx, s, g, batch);
            else launch<MI355_DTYPE_F16, MI355_DTYPE_F16, true>(ctx, s, g, batch);
        }
    }
    check_launch(ctx, "mi355_gemm(lp128)");
    return MI355_OK;
}

// ---- the 256 x 128 tile (MI = 4): 16-bit operands, at most one round of tiles (one workgroup per CU: 144 KiB of LDS) --------
bool gemm_lp256x128_supports(const mi355_gemm_desc &d, const void *a, const void *b, const void *c)
{
    if (d.dtype_ab != MI355_DTYPE_BF16 && d.dtype_ab != MI355_DTYPE_F16) return false;
    if (d.trans_a) return false;
    if ((int64_t)256 * std::max(d.lda, d.ldb) * 2 >= (1ll << 32)) return false;   // 256 tile rows: the per-lane DMA offsets of rows 128-255 are 32-bit too
    return gemm_lp128_supports(d, a, b, c);
}

int32_t launch_gemm_lp256x128(mi355_ctx *ctx, hipStream_t s, const mi355_gemm_desc &d, const void *a, const void *b, void *c)
{
    if (!gemm_lp256x128_supports(d, a, b, c))
        return fail(ctx, MI355_E_UNSUPPORTED, "lp256x128 GEMM: shape/layout not supported by this kernel");
    gemm_args g{};
    g.a = a; g.b = b; g.c = c;
    g.m = d.m; g.n = d.n; g.k = d.k;
    g.lda = d.lda; g.ldb = d.ldb; g.ldc = d.ldc;
    g.stride_a = d.stride_a; g.stride_b = d.stride_b; g.stride_c = d.stride_c;
    g.tiles_m = (uint32_t)((d.m + 255) / 256);
    g.tiles_n = (uint32_t)((d.n + BN - 1) / BN);
    g.group_m = 4;                        // 4 x 8 tiles of 256 x 128 = the 1024 x 1024 patch an XCD's 32 workgroups share
    g.split_k = 1;
    g.split_c_stride = 0;
    const uint32_t batch = (uint32_t)d.batch;
#define TALL(DT_, DC_) { if (d.trans_b) launch_ns<DT_, DC_, 3, !LP128_INSTREAM, false, 4>(ctx, s, g, batch); else launch_ns<DT_, DC_, 3, !LP128_INSTREAM, true, 4>(ctx, s, g, batch); }
    if (d.dtype_ab == MI355_DTYPE_BF16) {
        if (d.dtype_c == MI355_DTYPE_F32) TALL(MI355_DTYPE_BF16, MI355_DTYPE_F32) else TALL(MI355_DTYPE_BF16, MI355_DTYPE_BF16)
    } else {
        if (d.dtype_c == MI355_DTYPE_F32) TALL(MI355_DTYPE_F16, MI355_DTYPE_F32) else TALL(MI355_DTYPE_F16, MI355_DTYPE_F16)
    }
#undef TALL
    check_launch(ctx, "mi355_gemm(lp256x128)");
    return MI355_OK;
}

}  // namespace mi355
