// gemm_lp256p.hip -- the PERSISTENT form of gemm_lp256w4.hip (same 256x256 tile, 4 waves x 128x128, same ring
// of five 32 KiB LDS slots, same hand-pinned k-step; read that file's header first).
//
// What changes: one workgroup per CU walks several output tiles, and the K-tile stream does not stop at a tile
// boundary.  While the last two K-tiles of tile i are being multiplied, the DMA units being issued (always two
// K-tiles ahead) already belong to tile i+1, so the next tile's first fragments are in LDS before tile i's
// epilogue starts: no pipeline fill per tile, and the C stores of tile i drain under the MFMAs of tile i+1.
// Measured cost of the fill + drain + epilogue it hides: ~25k cycles per tile against 77k cycles of K loop at
// K = 2048 (config C5) and 154k at K = 4096.
//
// The epilogue needs LDS scratch while four of the five ring slots hold the next tile's first two K-tiles.
// The free one is the slot of the last B unit (dead since the last hand-over barrier); each wave stages in the
// 8 KiB of that slot that only ITS OWN later DMA pieces overwrite, so no barrier is needed around the epilogue:
// program order inside the wave (reads waited for before its next DMA issue) is the only ordering required.
// vmcnt: the epilogue's stores are older than the DMA loads that follow them and loads complete in order among
// themselves, so the counted wait at the next hand-over still proves the older loads have landed.
//
// Tile order: linear id L = blockIdx.x + round * gridDim.x over tiles x batch, passed through the same bijective
// XCD remap + grouped rasterisation -- exactly the tiles the non-persistent launch would have dispatched round by
// round (block b runs on XCD b % 8; speed only).
//
// Restrictions: as gemm_lp256w4.hip, plus K >= two K-tiles.
//
// STATUS (round 1): bit-identical to gemm_lp256w4.hip per tile (tests/test_gpu_gemm.py).  The first form was 2-11 % slower
// than the one-tile-per-workgroup kernel (f32 staging in two half-lane passes, per-lane 64-bit DMA pointers); with the
// scalar-base DMA addressing and, for 16-bit C, a staging image converted on the way into LDS (32 rows x 256 B = exactly the
// wave's 8 KiB, chunk index XOR-ed with the row; accumulators read element by element through `v_accvgpr_read` asm, because
// the compiler's bulk copy of all 256 accumulators spills inside the tile loop) it is now AHEAD whenever a launch has
// several rounds of short tiles -- interleaved A/B (tools/dev/p_vs_w4.py): K = 512 +10 %, 1024 +4 %, 2048 +1...4 %
// (config C5's shard 1 234-1 258 vs 1 202-1 211 TFLOP/s), 4096 +1 %, 8192 a tie, single-round launches -0.5...-2 %.
// MI355_GEMM_ALGO_AUTO therefore takes it for 16-bit operands when there are >= 512 tiles and K <= 4096, unless the
// strip split of a small leftover round applies (gemm.cpp).
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

// LDS-DMA with a wave-uniform 64-bit base in SGPRs + a constant 32-bit per-lane offset (see gemm_lp256w4.hip: hipcc turns
// per-lane pointers into one 64-bit vector add per piece in the K loop; worth 4 % there).
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
#define W4_ABL 0          // dev ablations: 1 no DMA, 2 no fragment reads, 4 no MFMA, 32 DMA re-reads K-tiles 0-2, 64 DMA off after K-tile 2
#endif

#define WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" W4_STR(n) ")" ::: "memory")
#ifndef W4_PF
#define W4_PF 0       // 1: L2 prefetch of K-tile t+4 (one global_load_dword per wave and K-tile).  Measured:
                      // +16 % on the DMA-only ablation, -1.5 % on the full kernel (12.7 M extra L2 requests for
                      // nothing: with MFMAs in the stream the DMA latency is already covered) => off.
#endif
#ifndef W4_EVEN
#define W4_EVEN 0     // 1: DMA pieces spread 4 per k-step (dev A/B)
#endif
#ifndef W4_NT_C
#define W4_NT_C 1     // 1: non-temporal C stores (+1 % at 8192^3, neutral at 4096^3) (keep A/B rather than C in the 256 MiB Infinity Cache)
#endif
#ifndef W4_VMW
#define W4_VMW (8 + W4_PF)   // outstanding VMEM instructions allowed at the K-tile hand-over
#endif
#define W4_STR_(x) #x
#define W4_STR(x) W4_STR_(x)
#define WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
template <int V> using IC = std::integral_constant<int, V>;

#ifdef P_TRACE
__device__ unsigned long long p_trace_buf[256 * 16];
__device__ __forceinline__ void p_stamp(int tid, int k) { if (tid == 0 && blockIdx.x < 256 && k < 16) p_trace_buf[blockIdx.x * 16 + k] = __builtin_amdgcn_s_memtime(); }
#define P_STAMP(k) p_stamp(tid, k)
#else
#define P_STAMP(k)
#endif

// BNN: B is row-major [K][N] instead of [N][K] (f32 and, since round 3, bf16 / f16: see gemm_lp256w4.hip).
template <int DT, int DT_C, bool BNN = false>
__global__ void __launch_bounds__(256)
gemm_lp256p_kernel(gemm_args g)
{
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    typedef typename lp<DT>::frag frag;
    // row-major B with 16-bit operands: the transposing-read image of gemm_lp256w4.hip ("BNN, bf16 / f16")
    constexpr bool BNN16 = BNN && DT != MI355_DTYPE_F32;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5, l31 = lane & 31;
    constexpr int ESZ = lp<DT>::ESZ;
    constexpr int BK = ROW_BYTES / ESZ;                 // 64 (16-bit) / 32 (f32) k-values per K-tile
    const int nk = (int)(g.k / BK);

    const uint32_t tiles = g.tiles_m * g.tiles_n;
    const uint32_t total = tiles * g.batch_count;

    // ---- where output tile L lives: per-lane DMA source pointers + scalar coordinates -------------------
    // DMA map (as gemm_lp256w4.hip): a unit is 32 pieces of 1 KiB (8 rows); this wave fills pieces wave*8 + j;
    // lane -> (row = piece*8 + lane/8, physical chunk c = lane%8), source chunk = c ^ ((row>>1)&7); two per-lane
    // pointers per operand (j parity), the (j>>1) step is a wave-uniform byte offset.
    const int sub = lane >> 3, c8 = lane & 7;
    // per tile: wave-uniform bases (first row of the tile; scalar registers); per lane, once: the byte offsets of its 16
    // bytes of every piece (full tiles only, so they do not depend on the tile)
    struct tile_src { const char *ua, *ub, *ubnn; int64_t m0, n0, batch; };
    auto locate = [&](uint32_t L) {
        tile_src t;
        const uint32_t R = xcd_remap(L, total);
        const uint32_t bi = R / tiles, tl = R - bi * tiles;
        uint32_t tm, tn;
        tile_coords(tl, g.tiles_m, g.tiles_n, g.group_m, tm, tn);
        t.m0 = (int64_t)tm * BM; t.n0 = (int64_t)tn * BN; t.batch = bi;
        const char *A = static_cast<const char *>(g.a) + (int64_t)bi * g.stride_a * ESZ;
        const char *B = static_cast<const char *>(g.b) + (int64_t)bi * g.stride_b * ESZ;
        t.ua = A + t.m0 * g.lda * ESZ;
        t.ub = B + t.n0 * g.ldb * ESZ;
        t.ubnn = B + (int64_t)(BNN16 ? wave * 16 : wave * 8) * g.ldb * ESZ + t.n0 * ESZ;
        return t;
    };
    uint32_t voff_a[8], voff_b[8], voff_bnn[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int r = wave * 64 + j * 8 + sub;
        const int q = c8 ^ ((r >> 1) & 7);
        voff_a[j] = (uint32_t)(r * g.lda * ESZ + q * 16);
        voff_b[j] = (uint32_t)(r * g.ldb * ESZ + q * 16);
        if constexpr (BNN16)    // piece wave*8 + j: block row a = 4 wave + j/2, blocks 4(j%2) + lane/16, row (lane%16)/4, chunk lane%4
            voff_bnn[j] = (uint32_t)(((j >> 1) * 4 + ((lane & 15) >> 2)) * g.ldb * ESZ + (j & 1) * 256 + (lane >> 4) * 64 + (lane & 3) * 16);
        else
            voff_bnn[j] = (uint32_t)(j * g.ldb * ESZ + lane * 16);
    }
    const int dst_piece = wave * 8 * 1024;                            // + j*1024 within the slot

    // ---- fragment read offsets: row*128 + ((2s+h) ^ f) * 16, f = (row>>1)&7 = (l31>>1)&7 for every tile row
    const int f = (l31 >> 1) & 7;
    const int rowoff_a = (wm * 128 + l31) * ROW_BYTES;
    const int rowoff_b = BNN16 ? wn * 4 * 256 + ((lane & 15) >> 2) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8
                         : BNN ? (wn * 128 + l31) * 4 : (wn * 128 + l31) * ROW_BYTES;

    f32x16 acc[4][4];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    zero_acc();

    frag fa[2][4], fb[2][4];
    tile_src iss;                                   // the tile whose K-tiles are being ISSUED (two ahead of the MFMAs)

    // fragment load order == order of first use by the next k-step's MFMAs (j outer, i inner)
    auto read_one = [&](auto buf, auto idx, const char *pa, const char *pb) {
        constexpr int BUF = decltype(buf)::value, R = decltype(idx)::value;
        if constexpr (BNN16) {
            if constexpr (R >= 1 && R <= 4) fa[BUF][R - 1] = *reinterpret_cast<const frag *>(pa + (R - 1) * 32 * ROW_BYTES);
            else {
                constexpr int JB = (R == 0) ? 0 : R - 4;       // column block JB: k 0..3 from block row a, k 4..7 from a + 1 (2 KiB on)
                typedef short s16x4 __attribute__((ext_vector_type(4)));
                typedef short s16x8 __attribute__((ext_vector_type(8)));
                const auto q = (__attribute__((address_space(3))) s16x4 *)(pb + JB * 256);
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(q);
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(q + 256);
                fb[BUF][JB] = __builtin_bit_cast(frag, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
        } else if constexpr (BNN) {
            if (R >= 1 && R <= 4) fa[BUF][R - 1] = *reinterpret_cast<const frag *>(pa + (R - 1) * 32 * ROW_BYTES);
            else {
                constexpr int JB = (R == 0) ? 0 : R - 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) fb[BUF][JB][e] = *reinterpret_cast<const float *>(pb + e * 1024 + JB * 128);
            }
        } else {
            if (R == 0) fb[BUF][0] = *reinterpret_cast<const frag *>(pb);
            else if (R <= 4) fa[BUF][R - 1] = *reinterpret_cast<const frag *>(pa + (R - 1) * 32 * ROW_BYTES);
            else fb[BUF][R - 4] = *reinterpret_cast<const frag *>(pb + (R - 4) * 32 * ROW_BYTES);
        }
    };
    auto dma_one = [&](auto is_b, auto jj, int64_t koff, char *base) {
        constexpr int J = decltype(jj)::value;
        if constexpr (BNN && decltype(is_b)::value) {
            glds16_s<J * 1024>(iss.ubnn + koff * g.ldb, voff_bnn[J], lds_addr_of(base));   // a K-tile is 32 rows of ldb elements here
        } else {
            glds16_s<J * 1024>((decltype(is_b)::value ? iss.ub : iss.ua) + koff, decltype(is_b)::value ? voff_b[J] : voff_a[J],
                               lds_addr_of(base));
        }
    };
    auto mfma_one = [&](auto buf, auto idx) {
        constexpr int BUF = decltype(buf)::value, I = decltype(idx)::value & 3, J = decltype(idx)::value >> 2;
        if constexpr (DT == MI355_DTYPE_F32) {
            constexpr int E = decltype(idx)::value >> 2;       // element-major: see gemm_lp256w4.hip
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int pair = (decltype(idx)::value & 3) * 4 + t, i = pair & 3, j = pair >> 2;
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[BUF][j][E], fa[BUF][i][E], acc[i][j], 0, 0, 0);
            }
        } else {
            acc[I][J] = lp<DT>::mfma(fb[BUF][J], fa[BUF][I], acc[I][J]);
        }
    };

#define W4_STEP_BODY(CUR, NXT, RMASK, DMASK, IS_B, J0)                                               \
    {                                                                                                \
        constexpr unsigned rmask_ = (RMASK), dmask_ = (DMASK);                                       \
        W4_GROUP(CUR, NXT, 0, rmask_, dmask_, IS_B, J0)  W4_GROUP(CUR, NXT, 1, rmask_, dmask_, IS_B, J0)   \
        W4_GROUP(CUR, NXT, 2, rmask_, dmask_, IS_B, J0)  W4_GROUP(CUR, NXT, 3, rmask_, dmask_, IS_B, J0)   \
        W4_GROUP(CUR, NXT, 4, rmask_, dmask_, IS_B, J0)  W4_GROUP(CUR, NXT, 5, rmask_, dmask_, IS_B, J0)   \
        W4_GROUP(CUR, NXT, 6, rmask_, dmask_, IS_B, J0)  W4_GROUP(CUR, NXT, 7, rmask_, dmask_, IS_B, J0)   \
        W4_GROUP(CUR, NXT, 8, rmask_, dmask_, IS_B, J0)  W4_GROUP(CUR, NXT, 9, rmask_, dmask_, IS_B, J0)   \
        W4_GROUP(CUR, NXT, 10, rmask_, dmask_, IS_B, J0) W4_GROUP(CUR, NXT, 11, rmask_, dmask_, IS_B, J0)  \
        W4_GROUP(CUR, NXT, 12, rmask_, dmask_, IS_B, J0) W4_GROUP(CUR, NXT, 13, rmask_, dmask_, IS_B, J0)  \
        W4_GROUP(CUR, NXT, 14, rmask_, dmask_, IS_B, J0) W4_GROUP(CUR, NXT, 15, rmask_, dmask_, IS_B, J0)  \
    }
#define W4_GROUP(CUR, NXT, IDX, rmask_, dmask_, IS_B, J0)                                            \
    mfma_one(IC<CUR>{}, IC<IDX>{});                                                                  \
    if constexpr ((rmask_ >> IDX) & 1u)                                                              \
        read_one(IC<NXT>{}, IC<__builtin_popcount(rmask_ & ((1u << IDX) - 1u))>{}, rd_a, rd_b);      \
    if constexpr ((dmask_ >> IDX) & 1u)                                                              \
        dma_one(IC<IS_B>{}, IC<J0 + __builtin_popcount(dmask_ & ((1u << IDX) - 1u))>{}, dma_koff, dma_base); \
    __builtin_amdgcn_sched_barrier(0);

    // ---- first tile of this workgroup: units 0..3 (its K-tiles 0 and 1), then the first fragments ---------
    uint32_t L = blockIdx.x;
    tile_src cur = locate(L);
    iss = cur;
    {
        const int64_t k0 = 0, k1 = ROW_BYTES;
        char *b0 = smem + dst_piece;
#define W4_PRO(IS_B, KOFF, SLOT)                                                                     \
        dma_one(IC<IS_B>{}, IC<0>{}, KOFF, b0 + SLOT * UNIT_BYTES); dma_one(IC<IS_B>{}, IC<1>{}, KOFF, b0 + SLOT * UNIT_BYTES); \
        dma_one(IC<IS_B>{}, IC<2>{}, KOFF, b0 + SLOT * UNIT_BYTES); dma_one(IC<IS_B>{}, IC<3>{}, KOFF, b0 + SLOT * UNIT_BYTES); \
        dma_one(IC<IS_B>{}, IC<4>{}, KOFF, b0 + SLOT * UNIT_BYTES); dma_one(IC<IS_B>{}, IC<5>{}, KOFF, b0 + SLOT * UNIT_BYTES); \
        dma_one(IC<IS_B>{}, IC<6>{}, KOFF, b0 + SLOT * UNIT_BYTES); dma_one(IC<IS_B>{}, IC<7>{}, KOFF, b0 + SLOT * UNIT_BYTES);
        W4_PRO(0, k0, 0) W4_PRO(1, k0, 1) W4_PRO(0, k1, 2) W4_PRO(1, k1, 3)
#undef W4_PRO
    }
    WAIT_VMCNT(16);                      // units 0, 1 landed (this wave's share)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    {
        const int x = (h ^ f) << 4;
        const char *rd_a = smem + rowoff_a + x, *rd_b = smem + UNIT_BYTES + rowoff_b + (BNN16 ? h * 4096 : BNN ? (4 * h) * 1024 : x);
        read_one(IC<0>{}, IC<0>{}, rd_a, rd_b); read_one(IC<0>{}, IC<1>{}, rd_a, rd_b);
        read_one(IC<0>{}, IC<2>{}, rd_a, rd_b); read_one(IC<0>{}, IC<3>{}, rd_a, rd_b);
        read_one(IC<0>{}, IC<4>{}, rd_a, rd_b); read_one(IC<0>{}, IC<5>{}, rd_a, rd_b);
        read_one(IC<0>{}, IC<6>{}, rd_a, rd_b); read_one(IC<0>{}, IC<7>{}, rd_a, rd_b);
    }
    __builtin_amdgcn_sched_barrier(0);

    int sa = 0;                          // ring byte offset of the A unit of the K-tile being multiplied
    int sb = UNIT_BYTES;                 // ... and of its B unit; the ring runs on across output tiles
    auto adv = [](int x, int n) { x += n * UNIT_BYTES; return x >= LDS_BYTES ? x - LDS_BYTES : x; };
    const int x1 = ((2 + h) ^ f) << 4, x2 = ((4 + h) ^ f) << 4, x3 = ((6 + h) ^ f) << 4, x0 = (h ^ f) << 4;
    const int y0 = BNN16 ? h * 4096 : BNN ? (4 * h) * 1024 : x0, y1 = BNN16 ? 8192 + h * 4096 : BNN ? (8 + 4 * h) * 1024 : x1,
              y2 = BNN16 ? 16384 + h * 4096 : BNN ? (16 + 4 * h) * 1024 : x2, y3 = BNN16 ? 24576 + h * 4096 : BNN ? (24 + 4 * h) * 1024 : x3;

    char *__restrict__ C = static_cast<char *>(g.c);
    constexpr int CSZ = (DT_C == MI355_DTYPE_F32) ? 4 : 2;

    int stamp_i = 1;
    P_STAMP(0);
    for (;;) {
        const uint32_t Lnext = L + gridDim.x;
        const bool has_next = Lnext < total;
        tile_src nxt = cur;
        if (has_next) nxt = locate(Lnext);
        int kbase = 0;                   // K-tile index of the issue side = t + 2 - kbase

        for (int t = 0; t < nk; ++t) {
            const int sa1 = adv(sa, 2), sb1 = adv(sb, 2);     // units of K-tile t+1
            const int s4 = adv(sa, 4);                        // unit 2t+4 -> slot of unit 2t-1
            const int s5 = sa;                                // unit 2t+5 -> slot of unit 2t
            if (t == nk - 2 && has_next) { iss = nxt; kbase = nk; }   // from here on the stream feeds the next tile
            const int64_t dma_koff = (int64_t)min(t + 2 - kbase, nk - 1) * ROW_BYTES;   // clamp: only without a next tile
            const char *rd_a, *rd_b;
            char *dma_base;
            // ---- k-step 0: reads of step 1 after MFMA 0-7, unit 2t+4 pieces 0-3 after MFMA 9,11,13,15
            rd_a = smem + sa + rowoff_a + x1; rd_b = smem + sb + rowoff_b + y1; dma_base = smem + s4 + dst_piece;
            W4_STEP_BODY(0, 1, 0x00FFu, 0xAA00u, 0, 0)
            // ---- k-step 1: reads of step 2, unit 2t+4 pieces 4-7
            rd_a = smem + sa + rowoff_a + x2; rd_b = smem + sb + rowoff_b + y2;
            W4_STEP_BODY(1, 0, 0x00FFu, 0xAA00u, 0, 4)
            // ---- k-step 2: reads of step 3, no DMA; then the K-tile hand-over
            rd_a = smem + sa + rowoff_a + x3; rd_b = smem + sb + rowoff_b + y3;
            W4_STEP_BODY(0, 1, 0x00FFu, 0x0000u, 0, 0)
            WAIT_VMCNT(8);                   // my share of the next K-tile landed; unit 2t+4 may still fly
            WAIT_LGKM0();                    // my reads of this K-tile are complete
            __builtin_amdgcn_s_barrier();    // BAR_t
            __builtin_amdgcn_sched_barrier(0);
            // ---- k-step 3: reads of step 0 of the next K-tile after even MFMAs, unit 2t+5 pieces 0-7 after odd ones
            rd_a = smem + sa1 + rowoff_a + x0; rd_b = smem + sb1 + rowoff_b + y0; dma_base = smem + s5 + dst_piece;
            W4_STEP_BODY(1, 0, 0x5555u, 0xAAAAu, 1, 0)
            sa = sa1;
            sb = sb1;
        }

        P_STAMP(stamp_i); ++stamp_i;           // loop end
        // ---- epilogue of this tile ----------------------------------------------------------------------
        // Lane (l31, h) holds, for every 32-row block i, row l31 and the column groups n = j*32 + 8q + 4h .. +3.
        // Each wave transposes through ITS 8 KiB of the dead B slot: the accumulators go to LDS as f32 straight
        // from the AGPRs (ds_write_b128 takes accumulator registers: no VGPR copies, so no register pressure
        // inside the tile loop), 16 rows x 512 B per pass (the lanes of the other row half idle for that pass);
        // rows are read back whole, converted to the output type on the way, and stored 16 B per lane.  The
        // staging image has no padding (it must fit 8 KiB): the 16-byte chunk index is XOR-swizzled with the row.
        {
            char *stage = smem + adv(sb, 3) + wave * 8192;          // slot of the last B unit, my DMA region of it
            auto swz = [](int r) { return (r & 15) ^ (((r & 15) << 1) & 16); };
            constexpr int LPR = 128 * CSZ / 16;                      // lanes per output row: 16 (16-bit C) / 32 (f32 C)
            constexpr int RPI = 64 / LPR;                            // rows per store instruction: 4 / 2
            const int64_t cbase = cur.batch * g.stride_c;
            char *crow = C + (cbase + (cur.m0 + wm * 128 + lane / LPR) * g.ldc + cur.n0 + wn * 128) * CSZ + (lane % LPR) * 16;
            const int64_t cstep = (int64_t)RPI * g.ldc * CSZ;
            const int srow = l31 & 15, sx = swz(srow);
#ifndef P_EPI16
#define P_EPI16 1
#endif
            if constexpr (CSZ == 2 && P_EPI16) {
                // 16-bit C: the block is converted on its way to LDS and staged as 32 rows x 256 B (exactly this wave's
                // 8 KiB, no padding: the 16-byte chunk index is XOR-ed with the row instead), read back as whole rows:
                // half the LDS traffic and instructions of the f32 staging below.
                (void)srow; (void)sx;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            char *d = stage + l31 * 256 + (((j * 4 + q) ^ (l31 & 15)) << 4) + 8 * h;
                            // explicit per-element accumulator reads: left to itself the compiler copies the whole 256-register
                            // accumulator into VGPRs in one go at the loop exit, which spills inside this tile loop
                            float x0, x1, x2, x3;
                            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x0) : "a"(acc[i][j][4 * q + 0]));
                            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x1) : "a"(acc[i][j][4 * q + 1]));
                            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x2) : "a"(acc[i][j][4 * q + 2]));
                            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x3) : "a"(acc[i][j][4 * q + 3]));
                            if constexpr (DT_C == MI355_DTYPE_BF16) {
                                bf16x4 v = {(__bf16)x0, (__bf16)x1, (__bf16)x2, (__bf16)x3};
                                *reinterpret_cast<bf16x4 *>(d) = v;
                            } else {
                                f16x4 v = {(_Float16)x0, (_Float16)x1, (_Float16)x2, (_Float16)x3};
                                *reinterpret_cast<f16x4 *>(d) = v;
                            }
                            if ((q & 1) == 1) __builtin_amdgcn_sched_barrier(0);   // keep the accumulator reads from piling up in VGPRs
                        }
                    WAIT_LGKM0();                                  // same-wave hand-over: DS ops of one wave execute in order
                    char *cdst = crow + (int64_t)(i * 32) * g.ldc * CSZ;
#pragma unroll
                    for (int it = 0; it < 32 / RPI; ++it) {
                        const int r = it * RPI + lane / LPR;
                        const u32x4 v = *reinterpret_cast<const u32x4 *>(stage + r * 256 + (((lane % LPR) ^ (r & 15)) << 4));
                        __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(cdst + it * cstep));
                    }
                    WAIT_LGKM0();                                  // staged rows are in registers before the next block overwrites them
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {
                    if ((l31 >> 4) == pass) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const int ch = (j * 8 + 2 * q + h) ^ sx;              // f32 column j*32 + 8q + 4h = chunk j*8 + 2q + h
                                f32x4 v = {acc[i][j][4 * q + 0], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                                *reinterpret_cast<f32x4 *>(stage + srow * 512 + (ch << 4)) = v;
                            }
                    }
                    WAIT_LGKM0();                                  // same-wave hand-over: DS ops of one wave execute in order
                    char *cdst = crow + (int64_t)(i * 32 + pass * 16) * g.ldc * CSZ;
#pragma unroll
                    for (int it = 0; it < 16 / RPI; ++it) {
                        const int r = it * RPI + lane / LPR;
                        const char *row = stage + r * 512;
                        if constexpr (DT_C == MI355_DTYPE_F32) {
                            const f32x4 v = *reinterpret_cast<const f32x4 *>(row + (((lane % LPR) ^ swz(r)) << 4));
                            __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(cdst + it * cstep));
                        } else {
                            const int p = lane % LPR;                                  // my 8 output columns = f32 chunks 2p, 2p+1
                            const f32x4 v0 = *reinterpret_cast<const f32x4 *>(row + (((2 * p) ^ swz(r)) << 4));
                            const f32x4 v1 = *reinterpret_cast<const f32x4 *>(row + (((2 * p + 1) ^ swz(r)) << 4));
                            if constexpr (DT_C == MI355_DTYPE_BF16) {
                                bf16x8 o = {(__bf16)v0[0], (__bf16)v0[1], (__bf16)v0[2], (__bf16)v0[3],
                                            (__bf16)v1[0], (__bf16)v1[1], (__bf16)v1[2], (__bf16)v1[3]};
                                __builtin_nontemporal_store(o, reinterpret_cast<bf16x8 *>(cdst + it * cstep));
                            } else {
                                f16x8 o = {(_Float16)v0[0], (_Float16)v0[1], (_Float16)v0[2], (_Float16)v0[3],
                                           (_Float16)v1[0], (_Float16)v1[1], (_Float16)v1[2], (_Float16)v1[3]};
                                __builtin_nontemporal_store(o, reinterpret_cast<f16x8 *>(cdst + it * cstep));
                            }
                        }
                    }
                    WAIT_LGKM0();                                  // staged rows are in registers before the next pass overwrites them
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        P_STAMP(stamp_i); ++stamp_i;           // epilogue end
        if (!has_next) break;
        zero_acc();
        cur = nxt;
        L = Lnext;
    }
#undef W4_STEP_BODY
#undef W4_GROUP
    WAIT_VMCNT(0);                       // drain the clamped tail DMA (and the last stores) before the workgroup retires
}

template <int DT, int DT_C, bool BNN = false>
void launch(mi355_ctx *ctx, hipStream_t s, const gemm_args &g, uint32_t batch)
{
    lds_opt_in(ctx, reinterpret_cast<const void *>(gemm_lp256p_kernel<DT, DT_C, BNN>), LDS_BYTES);
    // one workgroup per CU (LDS admits no more); fewer when there are fewer tiles than CUs
    const uint32_t total = g.tiles_m * g.tiles_n * batch;
    const uint32_t grid = std::min<uint32_t>(total, ctx->props.num_streaming_multiprocessors);
    hipLaunchKernelGGL((gemm_lp256p_kernel<DT, DT_C, BNN>), dim3(grid), dim3(256), LDS_BYTES, s, g);
}

}  // namespace

#ifdef P_TRACE
extern "C" __attribute__((visibility("default"))) int mi355_dev_p_trace(unsigned long long *host_out)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(p_trace_buf), sizeof(unsigned long long) * 256 * 16);
}
#endif

namespace mi355 {

bool gemm_lp256p_supports(const mi355_gemm_desc &d, const void *a, const void *b, const void *c)
{
    if (d.dtype_ab != MI355_DTYPE_BF16 && d.dtype_ab != MI355_DTYPE_F16 && d.dtype_ab != MI355_DTYPE_F32) return false;
    if (d.dtype_c != MI355_DTYPE_F32 && d.dtype_c != d.dtype_ab) return false;
    if (d.trans_a) return false;
    const int64_t esz = d.dtype_ab == MI355_DTYPE_F32 ? 4 : 2;
    const int64_t BK = ROW_BYTES / esz;
    if (d.k < 2 * BK || d.k % BK != 0) return false;              // the stream is two K-tiles deep
    const int64_t csz = d.dtype_c == MI355_DTYPE_F32 ? 4 : 2;      // the epilogue writes C in 16-byte pieces
    if (((d.ldc * csz) & 15) || ((d.stride_c * csz) & 15) || (reinterpret_cast<uintptr_t>(c) & 15u)) return false;
    if (d.m < BM || d.m % BM != 0 || d.n < BN || d.n % BN != 0) return false;
    const int64_t amask = 16 / esz - 1;                               // operand rows must be 16-byte aligned
    if ((d.lda & amask) || (d.ldb & amask) || (d.stride_a & amask) || (d.stride_b & amask)) return false;
    if ((reinterpret_cast<uintptr_t>(a) & 15u) || (reinterpret_cast<uintptr_t>(b) & 15u)) return false;
    const int64_t tiles = (d.m / BM) * (d.n / BN) * d.batch;
    if (tiles > 0x7FFFFFFF) return false;
    if ((int64_t)BM * std::max(d.lda, d.ldb) * esz >= (1ll << 32)) return false;   // per-lane DMA offsets are 32-bit
    return true;
}

int32_t launch_gemm_lp256p(mi355_ctx *ctx, hipStream_t s, const mi355_gemm_desc &d, const void *a, const void *b,
                            void *c)
{
    if (!gemm_lp256p_supports(d, a, b, c))
        return fail(ctx, MI355_E_UNSUPPORTED, "lp256p GEMM: shape/layout not supported by this kernel");
    gemm_args g{};
    g.a = a; g.b = b; g.c = c;
    g.m = d.m; g.n = d.n; g.k = d.k;
    g.lda = d.lda; g.ldb = d.ldb; g.ldc = d.ldc;
    g.stride_a = d.stride_a; g.stride_b = d.stride_b; g.stride_c = d.stride_c;
    g.tiles_m = (uint32_t)(d.m / BM);
    g.tiles_n = (uint32_t)(d.n / BN);
    g.group_m = W4_GROUP_M;
    g.batch_count = (uint32_t)d.batch;
    const uint32_t batch = (uint32_t)d.batch;
    if (d.dtype_ab == MI355_DTYPE_F32) {
        if (d.trans_b) launch<MI355_DTYPE_F32, MI355_DTYPE_F32, false>(ctx, s, g, batch);
        else launch<MI355_DTYPE_F32, MI355_DTYPE_F32, true>(ctx, s, g, batch);
    } else if (d.dtype_ab == MI355_DTYPE_BF16) {
        if (d.trans_b) {
            if (d.dtype_c == MI355_DTYPE_F32) launch<MI355_DTYPE_BF16, MI355_DTYPE_F32>(ctx, s, g, batch);
            else launch<MI355_DTYPE_BF16, MI355_DTYPE_BF16>(ctx, s, g, batch);
        } else {
            if (d.dtype_c == MI355_DTYPE_F32) launch<MI355_DTYPE_BF16, MI355_DTYPE_F32, true>(ctx, s, g, batch);
            else launch<MI355_DTYPE_BF16, MI355_DTYPE_BF16, true>(ctx, s, g, batch);
        }
    } else {
        if (d.trans_b) {
            if (d.dtype_c == MI355_DTYPE_F32) launch<MI355_DTYPE_F16, MI355_DTYPE_F32>(ctx, s, g, batch);
            else launch<MI355_DTYPE_F16, MI355_DTYPE_F16>(ctx, s, g, batch);
        } else {
            if (d.dtype_c == MI355_DTYPE_F32) launch<MI355_DTYPE_F16, MI355_DTYPE_F32, true>(ctx, s, g, batch);
            else launch<MI355_DTYPE_F16, MI355_DTYPE_F16, true>(ctx, s, g, batch);
        }
    }
    check_launch(ctx, "mi355_gemm(lp256p)");
    return MI355_OK;
}

}  // namespace mi355
