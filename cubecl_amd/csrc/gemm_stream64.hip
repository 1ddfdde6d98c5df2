// gemm_stream64.hip -- bf16 / f16 GEMM with 3 ... 64 rows (or columns): C[m][n] = sum_k A[m][k] * B[n][k], no split-K.
//
// Roofline: HBM.  The large operand is read once; the small one (<= 64 rows x K, <= 1 MiB at K = 8192) stays in L2 and is
// re-read by every workgroup.  The 128x128 kernel serves these shapes with split-K: few tiles, so K is cut into slices whose
// f32 partial slabs a second kernel folds (25 % extra traffic at 64 rows plus a launch).  Here a workgroup owns 32 rows of
// the large operand over the WHOLE K range instead -- 256 workgroups at N = 8192, one per CU, each streaming one contiguous
// 512 KiB region -- so nothing is split and nothing is folded.
//
//   * twelve waves: four MULTIPLYING waves, four loader waves for the small operand and four for the streamed one (two until
//     round 3: with a piece costing its issuing wave ~100 cycles, two waves x two pieces per K-tile left little slack against the
//     ~340 cycles a K-tile may take -- four waves, interleaved against two on cold operands: 16 x 28672 x 8192 85.8 -> 80.5 us,
//     32 x 14336 x 4096 -5 %, 128 MiB operands a tie; warm +5...10 % everywhere); loaders only issue LDS-DMA pieces
//     (`global_load_lds_dwordx4`, 1 KiB each) and wait for them.  Two LDS rings: 6-8 K-tiles of the
//     small operand (L2-resident: short look-ahead) and 24 K-tiles = 96 KiB of the streamed one (HBM latency x bandwidth).
//     Same 128-byte-row image and chunk swizzle as gemm_lp128.hip.
//   * ring slots change hands in GROUPS of 2 (MB = 2) or 4 (MB = 1) K-tiles, one s_barrier per group; a group's fragment reads
//     are all issued before its first MFMA.
//   * history: one ring for both operands (12-16 K-tiles) left only 40-48 KiB of the streamed operand in flight: 4.5 TB/s
//     at 64 x 8192 x 8192 (34-36 us), whatever the hand-over granularity.
//   * tried and dropped: the small operand straight from global memory into registers (32 rows x 32 bytes per load), the ring
//     carrying the streamed operand only -- 43 us: 32 partial lines per load instruction cost more than the ring traffic saved.
//   * the four multiplying waves split the K-tile's four k-steps: wave w multiplies k-step w of every K-tile (MB MFMAs of
//     32x32x16 per K-tile) into its own accumulators; the four partial sums meet once, at the end, through LDS, added in wave
//     order (deterministic).  That spreads the fragment reads evenly: MB + 1 `ds_read_b128` per wave and K-tile.
//
// Either operand may be the small one: with N <= 64 the roles swap and the output tile is stored transposed (C is at most
// 1 MiB; its stores do not matter).  Accumulation order differs from the other kernels (k-steps interleaved over four
// accumulators): within the parity tolerance, not bit-identical to them.
#include <algorithm>
#include <cstdlib>

#include "gemm_common.hpp"

using namespace mi355;

namespace {

#ifndef S64_SHIFT
#define S64_SHIFT 1   // dev: 0 = every workgroup starts at K-tile 0
#endif
#ifndef S64_ABL
#define S64_ABL 0   // dev, timing only: 1 = the small operand always from K-tile 0 (no L2 traffic for it), 2 = the streamed operand re-reads its first 8 K-tiles (no HBM traffic)
#endif
#ifndef S64_NLB
#define S64_NLB 4   // loader waves of the streamed operand: 4 (8 rows = one piece per K-tile each; round 3) or 2 (16 rows = two pieces each; rounds 1-2)
#endif
constexpr int NLB = S64_NLB, PPW = 4 / NLB;   // pieces per streamed-loader wave and K-tile
constexpr int ROW_BYTES = 128;            // one K-tile row: 64 x 16-bit
constexpr int BLK = 32 * ROW_BYTES;       // 32 rows of one operand: 4 KiB
constexpr int BN = 32;                    // streamed rows per workgroup

template <int DT> struct lp;
template <> struct lp<MI355_DTYPE_BF16> {
    typedef bf16x8 frag;
    static __device__ __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct lp<MI355_DTYPE_F16> {
    typedef f16x8 frag;
    static __device__ __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

struct stream_args {
    const void *small_;      // [small_rows][K]
    const void *big;         // [big_rows][K]
    void *out;
    int64_t ld_small, ld_big;                       // elements
    int64_t out_stride_small, out_stride_big;       // elements between consecutive small / big indices of the output
    int64_t stride_small, stride_big, stride_out;   // batch strides, elements
    int32_t small_rows, big_rows, k;
    int32_t dtype_c;
    int32_t stream_nt;       // the streamed operand is larger than the Infinity Cache could keep: non-temporal LDS-DMA pieces
    // K slices (round 5, NB > 1 forms): workgroup blockIdx.x = row block * slices + slice walks K-tiles [slice nk / slices,
    // (slice + 1) nk / slices); its f32 block goes to `partial` [batch][row block][slice][small 32 MB][streamed 32 NB] and the last
    // workgroup of a row block to arrive (ticket word per row block) adds the slices in slice order and writes the output
    int32_t slices;
    float *partial;
    unsigned int *tickets;
};

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
// NT: the piece carries the non-temporal hint.  Only ever the STREAMED operand's pieces (read exactly once), and only when that
// operand cannot stay in the 256 MiB Infinity Cache anyway (stream_args::stream_nt, set by the launcher): interleaved on cold
// operands 16 x 28672 x 8192 (470 MiB) 83.9 -> 76.6 us, 32 x 14336 x 4096 29.2 -> 27.8, 128 MiB operands a tie -- but a 128 MiB
// operand re-read by back-to-back launches loses the cache's help with the hint (25.5 -> 31.9 us), and on the small operand (re-read
// by every workgroup from L2) or on the tile kernels' operands the hint costs 15-25 % (profiles/r03_lds_dma_cache_policy.md).
template <bool NT = false>
__device__ __forceinline__ void glds16_s(const void *ubase_in, uint32_t voff, uint32_t lds_in)
{
    const uint64_t u = reinterpret_cast<uint64_t>(ubase_in);
    const uint64_t us = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32)) << 32) |
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u);
    const void *ubase = reinterpret_cast<const void *>(us);
    const uint32_t lds_byte_addr = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_in);
    // s_nop 4: gfx9 wants 5 wait states between a VALU write of an SGPR (the readfirstlanes above) and a VMEM read of it;
    // the compiler cannot pad inside inline asm (tools/hazard_scan.py, tests/test_abi_cpu.py)
    if constexpr (NT)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(voff), "s"(ubase), "s"(lds_byte_addr) : "memory");
    else
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" LDS_DMA_MOD ::"v"(voff), "s"(ubase), "s"(lds_byte_addr) : "memory");
}
__device__ __forceinline__ uint32_t lds_addr_of(const void *p)
{
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char *)p;
}

template <int N> __device__ __forceinline__ void wait_vmcnt()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

#ifndef S64_T1_G
#define S64_T1_G 3      // ring geometry of the two-workgroups-per-CU form with up to 32 small rows (G, SGA, SGB); round 3: (3, 3, 3) -- six
#define S64_T1_SGA 3    // K-tiles of look-ahead on both operands in 72 KiB -- for (4, 2, 2): 8 x 57344 x 4096 95 -> 78 us, 16 x 32000 x 4096 54 -> 50,
#define S64_T1_SGB 3    // 16 x 28672 x 8192 84.9 -> 81.5 on cold operands (profiles/r03_stream64_ring_geometry.md)
#endif
// Ring geometry per form: G = K-tiles per hand-over (one s_barrier per group), SGA / SGB = groups in the small / streamed ring.
// Ring geometry of the one-workgroup-per-CU forms, re-tuned in round 3 on COLD operands (rounds 1-2 tuned on one operand set,
// i.e. with the small operand always in the cache hierarchy).  What the cold runs want is look-ahead on the SMALL operand: the
// K stagger makes every workgroup the first of its XCD to touch 1/32 of the small operand's K-tiles, and with (G, SGA, SGB) =
// (4, 2, 6) its loaders ran four K-tiles ahead -- ~0.8 us against a ~2 us L2 miss.  Swept (profiles/r03_stream64_ring_geometry.md;
// product of the time = (4, 2, 6) / (2, 3, 12)): up to 32 rows (8, 2, 3) -- eight K-tiles ahead on both operands -- 16 x 8192 x 8192
// 30.7 -> 25.0 us cold, 22.9 -> 21.5 warm, 32 x 512 x 8192 20.8 -> 16.1; 33-64 rows (4, 3, 4): 64 x 8192 x 8192 34.5 -> 33.5 cold,
// 25.1 -> 23.5 warm.
#ifndef S64_O1_G
#define S64_O1_G 8      // up to 32 small rows: (G, SGA, SGB)
#define S64_O1_SGA 2
#define S64_O1_SGB 3
#endif
#ifndef S64_O2_G
#define S64_O2_G 4      // 33-64 small rows
#define S64_O2_SGA 3
#define S64_O2_SGB 4
#endif
#ifndef S64_T2_G
#define S64_T2_G 2      // two workgroups per CU, 33-64 small rows; round 3: (2, 3, 4) for (2, 2, 5) -- 48 x 28672 x 4096 68.6 -> 61.5 us,
#define S64_T2_SGA 3    // 40 x 16384 x 8192 56.9 -> 52.7, 64 x 32768 x 4096 78.8 -> 70.6, 64 x 28672 x 8192 and 64 x 14336 x 4096 ties (cold)
#define S64_T2_SGB 4
#endif
#ifndef S64_W2_G
#define S64_W2_G 2      // NB = 2 (64 streamed rows per workgroup, K in slices; one workgroup per CU), 33-64 small rows: (G, SGA, SGB):
#define S64_W2_SGA 3    // per K-tile 8 KiB + 8 KiB -- six K-tiles of the small operand ahead, 96 KiB of the streamed one in the ring
#define S64_W2_SGB 6
#endif
#ifndef S64_W1_G
#define S64_W1_G 4      // NB = 2, up to 32 small rows: per K-tile 4 KiB + 8 KiB
#define S64_W1_SGA 2
#define S64_W1_SGB 3
#endif
template <int MB, bool TWO, int NB = 1> struct ring_geom {
    static constexpr int G = NB == 2 ? (MB == 2 ? S64_W2_G : S64_W1_G) : (TWO && MB == 1) ? S64_T1_G : TWO ? S64_T2_G : MB == 2 ? S64_O2_G : S64_O1_G;
    static constexpr int SGA = NB == 2 ? (MB == 2 ? S64_W2_SGA : S64_W1_SGA) : (TWO && MB == 1) ? S64_T1_SGA : TWO ? S64_T2_SGA : (MB == 2 ? S64_O2_SGA : S64_O1_SGA);
    static constexpr int SGB = NB == 2 ? (MB == 2 ? S64_W2_SGB : S64_W1_SGB) : (TWO && MB == 1) ? S64_T1_SGB : TWO ? S64_T2_SGB : (MB == 2 ? S64_O2_SGB : S64_O1_SGB);
    static constexpr int LDS = SGA * G * MB * BLK + SGB * G * NB * BLK;      // small ring + streamed ring: 128-144 KiB, or 64-72 KiB x 2
};

// MB: 32-row blocks of the small operand (1: up to 32 rows, 2: up to 64).
// TWO: half-depth rings, two workgroups per CU -- for grids of more than one workgroup per CU, where a starting workgroup's
// empty ring and a finishing one's drain overlap the neighbour's streaming (16 x 28672 x 8192: 107 -> 83.5 us); with one
// workgroup per CU the deep rings win (64 x 8192 x 8192: 24.7 us against 33.9).
// NB (round 5): 32-row blocks of the STREAMED operand per workgroup.  With NB = 2 a workgroup owns 64 streamed rows over HALF of K
// (args.slices = 2): the small operand crosses the L2 -> LDS path half as often per streamed byte (at 64 x 8192 x 8192 each of
// the 256 workgroups pulled the whole 1 MiB of it: 1 MiB + 0.5 MiB streamed per CU), and a K-tile's LDS stage holds as much of the
// streamed operand as of the small one, so the ring keeps 80 KiB of it in flight where the NB = 1 form had 48 (measured bound of
// that form: ~4.1 TB/s whatever the HBM side did, profiles/r03_stream64_cold_ablation.txt).  The price is the slices' meeting
// in memory: f32 blocks + one ticket per row block, folded in slice order by the last workgroup to arrive (as gemm_nnrows.hip).
template <int DT, int MB, bool TWO, int NB = 1>
__global__ void __launch_bounds__(512 + 64 * NLB, TWO ? (NLB == 2 ? 5 : 6) : (NLB == 2 ? 2 : 3)) gemm_stream64_kernel(stream_args g)
{
    // Two rings.  The streamed operand needs DEPTH: ~25 GB/s per CU x ~2.5 us of HBM latency under load = ~64 KiB in flight
    // (with both operands in one 12-16 slot ring only 40-48 KiB of it were, and the kernel sat at 4.5 TB/s).  The small
    // operand is L2-resident and needs only a few K-tiles of look-ahead.  They cannot share loader waves: `vmcnt` retires in
    // order, so a wave waiting for a near small-operand piece would also wait for every far streamed piece it issued before.
    typedef ring_geom<MB, TWO, NB> RG;
    constexpr int G = RG::G, SGA = RG::SGA, SGB = RG::SGB;
    constexpr int SA = SGA * G, SB = SGB * G;
    constexpr int A_STAGE = MB * BLK;
    constexpr int B_STAGE = NB * BLK;
    constexpr int B_RING = SA * A_STAGE;                  // byte offset of the streamed ring
    constexpr int BNW = BN * NB;                          // streamed rows per workgroup
    static_assert(NLB == 4 || NB == 1, "the NB > 1 forms are written for four streamed-loader waves");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    typedef typename lp<DT>::frag frag;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);     // 0-3 multiply, 4-7 load the small operand, 8 .. 8 + NLB - 1 the streamed one
    const int w = wave_all & 3;
    const int h = lane >> 5, l31 = lane & 31;
    const int slices = NB == 1 ? 1 : g.slices;
    const int rb = NB == 1 ? (int)blockIdx.x : (int)(blockIdx.x / (uint32_t)slices), sl = NB == 1 ? 0 : (int)(blockIdx.x % (uint32_t)slices);
    const int64_t n0 = (int64_t)rb * BNW;
    const int nk_all = g.k / 64;
    const int tile0 = NB == 1 ? 0 : (int)((int64_t)sl * nk_all / slices);                     // this slice's K-tiles: [tile0, tile0 + nk)
    const int nk = NB == 1 ? nk_all : (int)((int64_t)(sl + 1) * nk_all / slices) - tile0;
    const int ng = (nk + G - 1) / G;
    // Workgroup j walks K starting at K-tile j mod nk and wraps.  Without it all 256 workgroups ask for the same 128-byte
    // column of their 32 rows at the same time: 8192 lines whose addresses differ only above bit 14 -- the same few HBM
    // channels -- and the kernel sat at 4.2 TB/s however deep the ring (with the streamed operand re-read from L2 instead:
    // 19.6 us, so everything but HBM fits in 60 % of the time).  Each output element still sums its K-tiles in one fixed order.
    const int shift = S64_SHIFT ? (int)((uint32_t)rb % (uint32_t)nk) : 0;
    auto phys = [&](int t) { const int p = t + shift; return tile0 + (p >= nk ? p - nk : p); };

    const char *small_ = static_cast<const char *>(g.small_) + (int64_t)blockIdx.y * g.stride_small * 2;
    const char *big = static_cast<const char *>(g.big) + (int64_t)blockIdx.y * g.stride_big * 2 + n0 * g.ld_big * 2;

    // a DMA piece is 1 KiB = 8 rows x 128 B; a lane fills 16 bytes: row 8p + lane / 8, physical chunk lane % 8 <- logical
    // chunk ^ swizzle (as gemm_lp128.hip)
    if (wave_all >= 8) {
        // ---- streamed operand: wave 8 fills rows 0-15, wave 9 rows 16-31 of every K-tile (two pieces each)
        const int half = wave_all - 8;       // (one of NLB loaders: rows half * 8 PPW ...)
        constexpr int PW = PPW * NB;         // pieces of a K-tile per loader wave
        uint32_t voff[PW];
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            const int r = (half * PW + j) * 8 + (lane >> 3);
            const int q = (lane & 7) ^ ((r >> 1) & 7);
            voff[j] = (uint32_t)(std::min<int64_t>(r, (int64_t)g.big_rows - n0 - 1) * g.ld_big * 2 + q * 16);   // rows past the edge re-read the last one
        }
        auto issue_group = [&](int gi) {
#pragma unroll
            for (int u = 0; u < G; ++u) {
                const int t = gi * G + u;
                if (t < nk) {
                    const uint32_t slot = lds_addr_of(smem + B_RING + (t % SB) * B_STAGE + half * PW * 1024);
#pragma unroll
                    for (int j = 0; j < PW; ++j) {
                        const char *src = big + (int64_t)(S64_ABL == 2 ? (t & 7) : phys(t)) * ROW_BYTES;
                        if (g.stream_nt) glds16_s<true>(src, voff[j], slot + j * 1024);      // (wave-uniform branch)
                        else glds16_s<false>(src, voff[j], slot + j * 1024);
                    }
                }
            }
        };
        for (int gi = 0; gi < std::min(SGB - 1, ng); ++gi) issue_group(gi);
        for (int gi = 0; gi < ng; ++gi) {
            // group gi has landed when at most the pieces of the SGB - 2 groups issued after it are outstanding (full groups
            // when they all exist); in the tail simply wait for everything
            if ((gi + SGB - 1) * G <= nk) wait_vmcnt<PW * G * (SGB - 2)>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();          // group gi is ready; everybody is done with group gi - 1
            __builtin_amdgcn_sched_barrier(0);
            if (gi + SGB - 1 < ng) issue_group(gi + SGB - 1);   // into the slots group gi - 1 just left
        }
        return;
    }
    if (wave_all >= 4) {
        // ---- small operand: wave 4 + w fills pieces w, w + 4 (MB = 2) of every K-tile's MB x 32 rows
        uint32_t voff[MB];
#pragma unroll
        for (int j = 0; j < MB; ++j) {
            const int r = (w + 4 * j) * 8 + (lane >> 3);
            const int q = (lane & 7) ^ ((r >> 1) & 7);
            voff[j] = (uint32_t)(std::min<int64_t>(r, (int64_t)g.small_rows - 1) * g.ld_small * 2 + q * 16);
        }
        auto issue_group = [&](int gi) {
#pragma unroll
            for (int u = 0; u < G; ++u) {
                const int t = gi * G + u;
                if (t < nk) {
                    const uint32_t slot = lds_addr_of(smem + (t % SA) * A_STAGE);
#pragma unroll
                    for (int j = 0; j < MB; ++j) glds16_s(small_ + (int64_t)(S64_ABL == 1 ? 0 : phys(t)) * ROW_BYTES, voff[j], slot + (w + 4 * j) * 1024);
                }
            }
        };
        for (int gi = 0; gi < std::min(SGA - 1, ng); ++gi) issue_group(gi);
        for (int gi = 0; gi < ng; ++gi) {
            if ((gi + SGA - 1) * G <= nk) wait_vmcnt<MB * G * (SGA - 2)>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (gi + SGA - 1 < ng) issue_group(gi + SGA - 1);
        }
        return;
    }

    // ---- multiplying waves: wave w takes k-step w of every K-tile -------------------------------------------------------
    f32x16 acc[NB][MB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][i][r] = 0.f;
    const int q = w * 2 + h;                                   // logical 16-byte chunk of my k-step for my lane half
    int off_s[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        const int row = i * 32 + l31;
        off_s[i] = row * ROW_BYTES + ((q ^ ((row >> 1) & 7)) << 4);
    }
    const int off_b = B_RING + l31 * ROW_BYTES + ((q ^ ((l31 >> 1) & 7)) << 4);
    // no scalar load in flight at the loop header as far as the compiler's wait-count bookkeeping knows (they share lgkmcnt
    // with the LDS and return out of order: one of them pending turns the loop's counted LDS waits into drains)
    __builtin_amdgcn_s_waitcnt(0xC07F);                        // lgkmcnt(0)
    __builtin_amdgcn_sched_barrier(0);
    // one group: all its fragment reads first, then its MFMAs -- one LDS latency per group instead of one per K-tile.  Whole
    // groups run without a branch (a branch around a read makes the compiler drain the LDS queue before every later read);
    // only the last, partial group tests its K-tiles.
    auto group = [&](int gi, auto whole_c) {
        constexpr bool WHOLE = decltype(whole_c)::value;
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        frag bfq[G][NB], afq[G][MB];
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const int t = gi * G + u;
            if (WHOLE || t < nk) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) bfq[u][nb] = *reinterpret_cast<const frag *>(smem + (t % SB) * B_STAGE + nb * BLK + off_b);
#pragma unroll
                for (int i = 0; i < MB; ++i) afq[u][i] = *reinterpret_cast<const frag *>(smem + (t % SA) * A_STAGE + off_s[i]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < G; ++u) {
            if (WHOLE || gi * G + u < nk) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int i = 0; i < MB; ++i) acc[nb][i] = lp<DT>::mfma(bfq[u][nb], afq[u][i], acc[nb][i]);   // operands swapped: a lane owns 4 consecutive streamed columns per quad
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // (the MFMAs need the fragments, so every ds_read of the group has completed before this wave reaches the next barrier)
    };
    const int ng_whole = nk / G;
    for (int gi = 0; gi < ng_whole; ++gi) group(gi, std::true_type{});
    if (ng_whole < ng) group(ng_whole, std::false_type{});

    // ---- the four k-step partials meet in LDS (the ring is dead), added in wave order -----------------------------------
    __syncthreads();                                            // loaders have left; the four of us are done reading the ring
    // Row pitch 33 floats (round 6): with 32 the 32 lanes of a half-wave -- rows l31 = 0..31, the same column -- stored to addresses 128
    // bytes apart = ONE bank, 32-way, 16 x NB x MB times per wave: rocprofv3 counted 3 712 conflict cycles per CU of the 64-row form's
    // 8 778 LDS-array cycles (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.42; 0.17 on the 32-row form), every one of them here -- the
    // K loop's fragment reads are conflict free (profiles/r06_stream64_pmc.txt).
#ifndef S64_PART_PITCH
#define S64_PART_PITCH 33
#endif
    constexpr int PP = S64_PART_PITCH;
    float *part = reinterpret_cast<float *>(smem);              // [wave][NB][MB][row 32][pitch 33], 4 x NB x MB x 4.1 KiB
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    part[(((w * NB + nb) * MB + i) * 32 + l31) * PP + 8 * qd + 4 * h + r] = acc[nb][i][4 * qd + r];   // lane (l31, h): row l31, cols 8 qd + 4 h + r
    __syncthreads();
    char *out = static_cast<char *>(g.out);
    auto wave_sum = [&](int blk, int row, int col) {             // the four k-step partials of one element, in wave order
        float v = part[((0 * NB * MB + blk) * 32 + row) * PP + col];
#pragma unroll
        for (int ww = 1; ww < 4; ++ww) v += part[((ww * NB * MB + blk) * 32 + row) * PP + col];
        return v;
    };
    auto store_out = [&](int64_t sm, int64_t bg, float v) {
        const int64_t o = (int64_t)blockIdx.y * g.stride_out + sm * g.out_stride_small + bg * g.out_stride_big;
        if (g.dtype_c == MI355_DTYPE_F32) reinterpret_cast<float *>(out)[o] = v;
        else if (g.dtype_c == MI355_DTYPE_BF16) reinterpret_cast<uint16_t *>(out)[o] = f32_to_bf16_rne(v);
        else reinterpret_cast<uint16_t *>(out)[o] = f32_to_f16_rne(v);
    };
    if (NB == 1 || slices == 1) {
        for (int e = tid; e < NB * MB * 32 * 32; e += 256) {      // tid < 256 here: the loader waves have left
            const int blk = e / 1024, nb = blk / MB, i = blk % MB, row = (e >> 5) & 31, col = e & 31;
            const int64_t sm = i * 32 + row, bg = n0 + nb * 32 + col;
            if (sm >= g.small_rows || bg >= g.big_rows) continue;
            store_out(sm, bg, wave_sum(blk, row, col));
        }
        return;
    }
    if constexpr (NB > 1) {
        // ---- K slices meet in memory: [batch][row block][slice][blk = nb * MB + i][row 32][col 32] f32 ---------------------------
        // Hand-off as in reduce.hip / gemm_nnrows.hip: agent-scope 8-byte atomic stores (write-through), drained per wave, then the
        // row block's ticket; agent-scope loads in the folding workgroup; no fences (a fence per workgroup = an L2 write-back).
        typedef __attribute__((address_space(1))) unsigned long long gu64;
        typedef __attribute__((address_space(1))) unsigned int gu32;
        constexpr int BLOCK_F = NB * MB * 1024;                  // floats of one workgroup's block
        const int64_t nblocks = (g.big_rows + BNW - 1) / BNW;
        float *slab0 = g.partial + (((int64_t)blockIdx.y * nblocks + rb) * slices) * BLOCK_F;
        gu64 *mine = (gu64 *)reinterpret_cast<unsigned long long *>(slab0 + (int64_t)sl * BLOCK_F);
        for (int e2 = tid; e2 < BLOCK_F / 2; e2 += 256) {
            const int e = e2 * 2, blk = e / 1024, row = (e >> 5) & 31, col = e & 31;
            const float v0 = wave_sum(blk, row, col), v1 = wave_sum(blk, row, col + 1);
            __hip_atomic_store(mine + e2, ((unsigned long long)__float_as_uint(v1) << 32) | __float_as_uint(v0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // my partials have left the wave ...
        __syncthreads();                                          // ... every wave's have, before the ticket
        __shared__ unsigned int is_last;
        unsigned int *ticket = g.tickets + (int64_t)blockIdx.y * nblocks + rb;
        if (tid == 0) is_last = __hip_atomic_fetch_add((gu32 *)ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned int)slices - 1u ? 1u : 0u;
        __syncthreads();
        if (!is_last) return;
        // the last workgroup: every slice's block in slice order (its own comes back from L2), all loads of a pass in flight
        gu64 *all = (gu64 *)reinterpret_cast<unsigned long long *>(slab0);
        constexpr int PER = BLOCK_F / 2 / 256;                   // 8-byte pairs per thread: 8 (NB x MB = 4), 4 (= 2)
        for (int p0 = 0; p0 < PER; p0 += 4) {
            unsigned long long wv[4][4];                          // [pair][slice]: up to 4 slices
#pragma unroll
            for (int pp = 0; pp < 4; ++pp)
#pragma unroll
                for (int ss = 0; ss < 4; ++ss)
                    wv[pp][ss] = __hip_atomic_load(all + (int64_t)(ss < slices ? ss : slices - 1) * (BLOCK_F / 2) + (p0 + pp) * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
                float v0 = 0.f, v1 = 0.f;
#pragma unroll
                for (int ss = 0; ss < 4; ++ss)
                    if (ss < slices) { v0 += __uint_as_float((uint32_t)wv[pp][ss]); v1 += __uint_as_float((uint32_t)(wv[pp][ss] >> 32)); }
                const int e = ((p0 + pp) * 256 + tid) * 2, blk = e / 1024, nb = blk / MB, i = blk % MB, row = (e >> 5) & 31, col = e & 31;
                const int64_t sm = i * 32 + row, bg = n0 + nb * 32 + col;
                if (sm < g.small_rows && bg < g.big_rows) store_out(sm, bg, v0);
                if (sm < g.small_rows && bg + 1 < g.big_rows) store_out(sm, bg + 1, v1);
            }
        }
        if (tid == 0) __hip_atomic_store((gu32 *)ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next call
    }
}

template <int DT, int MB, bool TWO>
void launch_form(mi355_ctx *ctx, hipStream_t s, const stream_args &g, uint32_t batch)
{
    constexpr int LDS = ring_geom<MB, TWO>::LDS;
    lds_opt_in(ctx, reinterpret_cast<const void *>(gemm_stream64_kernel<DT, MB, TWO>), LDS);
    hipLaunchKernelGGL((gemm_stream64_kernel<DT, MB, TWO>), dim3((uint32_t)((g.big_rows + BN - 1) / BN), batch), dim3(512 + 64 * NLB), LDS, s, g);
}

// the NB = 2 form: 64 streamed rows per workgroup, K in g.slices slices (one workgroup per CU: the deep rings)
template <int DT, int MB>
void launch_wide(mi355_ctx *ctx, hipStream_t s, const stream_args &g, uint32_t batch)
{
    constexpr int LDS = ring_geom<MB, false, 2>::LDS;
    lds_opt_in(ctx, reinterpret_cast<const void *>(gemm_stream64_kernel<DT, MB, false, 2>), LDS);
    const uint32_t nblocks = (uint32_t)((g.big_rows + 2 * BN - 1) / (2 * BN));
    hipLaunchKernelGGL((gemm_stream64_kernel<DT, MB, false, 2>), dim3(nblocks * (uint32_t)g.slices, batch), dim3(512 + 64 * NLB), LDS, s, g);
}

template <int DT, int MB>
void launch_one(mi355_ctx *ctx, hipStream_t s, const stream_args &g, uint32_t batch)
{
    if (g.slices > 1) return launch_wide<DT, MB>(ctx, s, g, batch);
    const uint64_t wgs = (uint64_t)((g.big_rows + BN - 1) / BN) * batch;
    if (wgs > (uint64_t)ctx->props.num_streaming_multiprocessors) launch_form<DT, MB, true>(ctx, s, g, batch);
    else launch_form<DT, MB, false>(ctx, s, g, batch);
}

}  // namespace

namespace mi355 {

constexpr int SCRATCH_STREAM64 = 8;     // library scratch kind: the K-slice blocks of the NB = 2 form (gemm_common.hpp lists the others)

// The cut of a launch: 0 = the NB = 1 form (32 streamed rows per workgroup over the whole K), else the number of K slices of the
// NB = 2 form (64 streamed rows per workgroup).  A pure function of the descriptor and the CU count.  Taken where the NB = 1 form
// runs one workgroup per CU or less over a long K -- the grid whose small operand crosses the L2 -> LDS path once per workgroup
// (profiles/r05_stream64_wide.txt).  MI355_S64_WIDE=0 / 1 (dev) forces the choice where the form is possible at all.
int stream64_slices(const mi355_gemm_desc &d, int cus)
{
    const int64_t small_rows = std::min(d.m, d.n), big_rows = std::max(d.m, d.n), nk = d.k / 64;
    const int64_t wgs1 = (big_rows + BN - 1) / BN * d.batch, nblocks = (big_rows + 2 * BN - 1) / (2 * BN);
    const bool possible = big_rows >= 2 * BN && nk >= 4 && nblocks * d.batch <= 448;     // one ticket word per row block: 496 per stream
    if (!possible) return 0;
    static const int forced = [] { const char *e = getenv("MI355_S64_WIDE"); return e ? atoi(e) : -1; }();
    if (forced == 0) return 0;
    if (forced > 0) return forced >= 2 && forced <= 4 ? forced : 2;
    // Measured (cold, us, NB = 1 / NB = 2 with two slices): 64 x 8192 x 8192 36.4 / 35.1, 8192 x 64 x 8192 36.2 / 34.9, 64 x 7168 x 8192
    // 35.4 / 33.7, 64 x 8192 x 16384 58.9 / 50.4; not taken: 48 x 8192 x 8192 32.1 / 33.1, 40 rows 29.7 / 32.5, 64 x 8192 x 4096 23.2 / 23.5,
    // up to 32 rows +3 us, four slices +6 us everywhere.  (The form was built on the reading that the small operand's trips through
    // L2 -> LDS bound the 64-row case; halving them bought 4 % at K = 8192 -- that reading is wrong, see the profile note.)
    return (small_rows >= 56 && wgs1 <= cus && wgs1 > cus / 2 && nk >= 128) ? 2 : 0;
}

// A [M][K], B [N][K] K-contiguous 16-bit (f32: the form in gemm_stream64_f32.hip), min(M, N) <= 64, K a multiple of 64, 16-byte aligned rows.
bool gemm_stream64_supports(const mi355_gemm_desc &d, const void *a, const void *b, const void *c)
{
    (void)c;
    if (d.dtype_ab == MI355_DTYPE_F32) return gemm_stream64_f32_supports(d, a, b, c);      // the f32 form: gemm_stream64_f32.hip
    if (d.dtype_ab != MI355_DTYPE_BF16 && d.dtype_ab != MI355_DTYPE_F16) return false;
    if (d.dtype_c != MI355_DTYPE_F32 && d.dtype_c != MI355_DTYPE_BF16 && d.dtype_c != MI355_DTYPE_F16) return false;
    if (d.trans_a || !d.trans_b) return false;
    if (d.m <= 0 || d.n <= 0 || d.k < 64 || (d.k & 63) || d.k > 0x7FFFFFC0) return false;
    if (std::min(d.m, d.n) > 64) return false;
    if (d.m > 0x7FFFFFFF || d.n > 0x7FFFFFFF || d.batch < 1 || d.batch > 65535) return false;
    if ((d.lda & 7) || (d.ldb & 7) || (d.stride_a & 7) || (d.stride_b & 7)) return false;
    if ((reinterpret_cast<uintptr_t>(a) & 15u) || (reinterpret_cast<uintptr_t>(b) & 15u)) return false;
    if ((int64_t)64 * std::max(d.lda, d.ldb) * 2 >= (1ll << 32)) return false;     // per-lane DMA offsets are 32-bit
    return true;
}

int32_t launch_gemm_stream64(mi355_ctx *ctx, hipStream_t s, const mi355_gemm_desc &d, const void *a, const void *b, void *c)
{
    if (d.dtype_ab == MI355_DTYPE_F32) return launch_gemm_stream64_f32(ctx, s, d, a, b, c);
    if (!gemm_stream64_supports(d, a, b, c)) return fail(ctx, MI355_E_UNSUPPORTED, "mi355_gemm: the 64-row streaming kernel does not take this descriptor");
    stream_args g{};
    const bool a_small = d.m <= d.n;
    g.small_ = a_small ? a : b;
    g.big = a_small ? b : a;
    g.out = c;
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
    g.dtype_c = d.dtype_c;
    // streamed bytes of the whole launch against what the 256 MiB Infinity Cache could still hold next to everything else
    g.stream_nt = (int64_t)g.big_rows * d.k * 2 * std::max<int64_t>(d.batch, 1) > (192ll << 20) ? 1 : 0;
    const bool two = g.small_rows > 32;
    const uint32_t batch = (uint32_t)d.batch;
    // K slices (the NB = 2 form): scratch for the slices' f32 blocks + the stream's ticket words; inside a capture window that does
    // not have them yet the launch falls back to the NB = 1 form, which needs neither
    const int cus = ctx->props.num_streaming_multiprocessors > 0 ? ctx->props.num_streaming_multiprocessors : 256;
    g.slices = stream64_slices(d, cus);
    if (g.slices > 1) {
        const int64_t nblocks = (g.big_rows + 2 * BN - 1) / (2 * BN);
        const size_t bytes = (size_t)batch * nblocks * g.slices * 2 * (two ? 2 : 1) * 1024 * sizeof(float);
        void *part = nullptr;
        const bool can_create = !ctx->capturing || (ctx->ticket_buf && !ctx->tickets_dirty && ctx->scratch.count({s, SCRATCH_STREAM64}) &&
                                                    ctx->scratch[{s, SCRATCH_STREAM64}].second >= bytes);
        if (can_create && scratch_get(ctx, s, SCRATCH_STREAM64, bytes, &part) == MI355_OK && strip_tickets_for_stream(ctx, s, &g.tickets) == MI355_OK)
            g.partial = static_cast<float *>(part);
        else g.slices = 0;
    }
    if (g.slices <= 1) g.slices = 1;
    if (d.dtype_ab == MI355_DTYPE_BF16) {
        if (two) launch_one<MI355_DTYPE_BF16, 2>(ctx, s, g, batch);
        else launch_one<MI355_DTYPE_BF16, 1>(ctx, s, g, batch);
    } else {
        if (two) launch_one<MI355_DTYPE_F16, 2>(ctx, s, g, batch);
        else launch_one<MI355_DTYPE_F16, 1>(ctx, s, g, batch);
    }
    if (g.slices > 1 && hipPeekAtLastError() != hipSuccess) strip_tickets_mark_dirty(ctx);   // a refused launch never resets its tickets
    check_launch(ctx, "mi355_gemm(stream64)");
    return MI355_OK;
}

}  // namespace mi355
