// gemm_lp256q.hip -- the persistent 256x256 kernel (gemm_lp256p.hip; read that file and gemm_lp256w4.hip first) with the
// C STORES DRIPPED INTO THE NEXT TILE'S K LOOP.  16-bit operands and 16-bit C only (config C5: batched 2048^3 bf16).
//
// What was measured on the persistent kernel (profiles/r01_power_ablation.md section 8): compiling the C stores out gains
// 10 % on the C5 shard, and neither the LDS transposition nor a chip-wide burst explains it -- the cost is per CU: 128 KiB
// of stores per tile leave a CU at ~16 B per cycle (~8k cycles) while its matrix pipe idles, and because CDNA4's vmcnt
// counts stores IN ORDER with the LDS-DMA loads, the first hand-over waits of the next tile sit behind them.  The only
// place the finished tile can wait while the next one accumulates is the register file (128 KiB of bf16 = 128 VGPRs per
// lane; LDS is full with the operand ring), and the budget is 256 architectural VGPRs next to the 256 accumulators:
//   * row blocks i = 0..2 of a wave's 128x128 block (96 of the 128 packed registers) are converted to 16-bit straight out
//     of the accumulators during the LAST k-step of the tile (drain of block b follows MFMA b+2: the matrix pipe keeps
//     issuing) and then leave, D stores per K-tile, during the next tile's K loop: 24 stores of 1 KiB per wave, each
//     writing 8 whole 128-byte lines after an in-register transposition (v_permlane32_swap + quad-permute DPP, no LDS).
//   * row block i = 3 goes through the LDS transposition of gemm_lp256p.hip at the tile boundary (8 stores per wave):
//     a quarter of the old epilogue.  Holding it too would need 128 + ~125 VGPRs; hipcc spills there.
//   * the accumulators are never zeroed: the first k-step of a tile multiplies into a literal zero C operand
//     (`v_mfma ... , 0`), which also removes the 256 v_accvgpr_write per tile of the persistent kernel.
// vmcnt discipline.  Loads and stores of a wave retire in issue order on this target (one counter, which is what makes a
// counted wait after mixed traffic meaningful -- LLVM relies on the same property for gfx9).  A hand-over needs "every DMA
// piece up to unit 2t+3 has landed"; what may still fly is whatever was issued after that unit: the 8 pieces of unit 2t+4
// plus the stores issued between the two units.  The dripped stores are therefore placed at the END of a K-tile (after its
// last DMA piece): the next hand-over counts them as allowed-in-flight (vmcnt(8 + D)) and only the hand-over after that
// (1.75 K-tiles = ~4000 cycles later) needs them gone.  Every K-tile body below is instantiated with its exact count.
//
// Restrictions: as gemm_lp256p.hip, plus 16-bit C of the operand type, lda == ldb (B stored [N][K]) and K >= (24 / D + 3) K-tiles.
//
// BNN (round 3): B row-major [K][N], the layout TensorHandle::new_contiguous gives a rhs.  Image, DMA map and the two
// ds_read_b64_tr_b16 per fragment are those of gemm_lp256w4.hip ("BNN, bf16 / f16"); tiles are full here, so the piece
// offsets are scalar (wave-uniform base arithmetic) and B's whole per-lane part is ONE register -- the held tile leaves
// room for no more.
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
constexpr int NHELD = 24;                         // dripped stores per wave and tile: 3 row blocks x 4 column blocks x 2

template <int DT> struct lp;
template <> struct lp<MI355_DTYPE_BF16> {
    typedef bf16x8 frag;
    static __device__ __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c)
    { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ uint32_t pack2(float x, float y)
    {
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        const bf16x2 v = {(__bf16)x, (__bf16)y};
        return __builtin_bit_cast(uint32_t, v);
    }
};
template <> struct lp<MI355_DTYPE_F16> {
    typedef f16x8 frag;
    static __device__ __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c)
    { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
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

#ifndef Q_ABL
#define Q_ABL 0           // dev ablations: 1 no dripped stores (timing only), 2 no boundary stores (timing only)
#endif
#ifndef Q_NT
#define Q_NT 1            // 1: non-temporal dripped stores, 0: plain
#endif
#ifndef Q_WAIT_EXTRA
#define Q_WAIT_EXTRA 0    // dev, timing only (unsafe): this many more operations may fly at every drip-phase hand-over
#endif
#ifndef Q_SLICED
#define Q_SLICED 1        // 1: a segment's transposition is slotted between the MFMAs of k-step 2; 0: in one piece in front of its first store
#endif
#ifndef Q_LDSDRAIN
#define Q_LDSDRAIN 1      // 1: row block 3 is packed into the staging image during the last k-step; 0: at the tile boundary
#endif
#ifndef Q_STORE_AT
#define Q_STORE_AT 2      // 2 = dripped stores at the end of k-step 2, in front of the hand-over (measured +4 % TFLOP/s per GHz over 3); 3 = behind the last DMA piece of k-step 3
#endif
#ifndef Q_STORE_FORM
#define Q_STORE_FORM 0    // dev: 0 builtin store (64-bit per-lane address), 1 inline asm: wave-uniform base in SGPRs + 32-bit lane offset
#endif
#ifndef Q_DUMMY
#define Q_DUMMY 0         // dev, timing only: every dripped store goes to the same 32 KiB of C (cache resident)
#endif
#ifndef Q_FORCE_D
#define Q_FORCE_D 0       // dev: stores per K-tile regardless of K
#endif
#define Q_STR_(x) #x
#define Q_STR(x) Q_STR_(x)
#define WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")   /* (n may depend on a template parameter) */
#define WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
template <int V> using IC = std::integral_constant<int, V>;
// Dev timing trace (-DQ_TRACE, never in the product library): shader-clock stamps of wave 0 of every workgroup for its first
// 8 tiles -- {K loop entered, K loop left (boundary begins), boundary left} -- read back with mi355_dev_q_trace
// (tools/dev/q_trace.py: K-loop cycles per K-tile against the 2 048-cycle matrix-pipe floor, boundary cycles per tile).
#ifdef Q_TRACE
__device__ unsigned long long q_trace_buf[256 * 32];
#define Q_STAMP(slot) do { if (tid == 0 && blockIdx.x < 256 && qt_tile < 8) q_trace_buf[blockIdx.x * 32 + qt_tile * 3 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define Q_STAMP(slot)
#endif
// A lane-constant value the compiler must re-derive where it is used: without this, hipcc hoists the 16 + 8 per-lane LDS
// addresses of the boundary staging image to kernel entry, runs out of registers beside the held tile, spills them, and
// every reload in the tile's last K-tile carries an s_waitcnt vmcnt(0) that drains the LDS-DMA stream (guide: "a
// lane-constant address hoisted to kernel entry is spilled around the tile loop ... recompute per block").
__device__ __forceinline__ uint32_t opaque(uint32_t x) { asm volatile("" : "+v"(x)); return x; }

// D = dripped stores per K-tile (1, 2, 4 or 8): the 24 held stores leave during the first 24 / D K-tiles of the next tile
template <int DT, int D, bool BNN = false>
__global__ void __launch_bounds__(256)
gemm_lp256q_kernel(gemm_args g)
{
    static_assert(NHELD % D == 0, "the held stores must split evenly over K-tiles");
    constexpr int S = NHELD / D;                    // K-tiles of the drip phase
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    typedef typename lp<DT>::frag frag;
    constexpr int ESZ = 2, CSZ = 2;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5, l31 = lane & 31;
    const int nk = (int)(g.k / 64);

    const uint32_t tiles = g.tiles_m * g.tiles_n;
    const uint32_t total = tiles * g.batch_count;

    struct tile_src { const char *ua, *ub; int64_t m0, n0, batch; };
    auto locate = [&](uint32_t L) {
        tile_src t;
        const uint32_t R = xcd_remap(L, total);
        const uint32_t bi = R / tiles, tl = R - bi * tiles;
        uint32_t tm, tn;
        tile_coords(tl, g.tiles_m, g.tiles_n, g.group_m, tm, tn);
        t.m0 = (int64_t)tm * BM; t.n0 = (int64_t)tn * BN; t.batch = bi;
        t.ua = static_cast<const char *>(g.a) + ((int64_t)bi * g.stride_a + t.m0 * g.lda) * ESZ;
        if constexpr (BNN)   // column n0 of k-row 16 wave: this wave's pieces are blocks rows a = 4 wave .. 4 wave + 3 (k-rows 16 wave .. +15)
            t.ub = static_cast<const char *>(g.b) + ((int64_t)bi * g.stride_b + t.n0 + (int64_t)(wave * 16) * g.ldb) * ESZ;
        else
            t.ub = static_cast<const char *>(g.b) + ((int64_t)bi * g.stride_b + t.n0 * g.ldb) * ESZ;
        return t;
    };
    // DMA map (gemm_lp256w4.hip): a unit is 32 pieces of 1 KiB (8 rows); this wave fills pieces wave*8 + j;
    // lane -> (row = piece*8 + lane/8, physical chunk c = lane%8), source chunk = c ^ ((row>>1)&7)
    const int sub = lane >> 3, c8 = lane & 7;
    uint32_t voff[8];                   // lda == ldb (supports()): one set of per-lane offsets serves both operands -- 8 VGPRs
#pragma unroll                          // fewer, which is what keeps the held tile + the K loop inside 256 registers
    for (int j = 0; j < 8; ++j) {
        const int r = wave * 64 + j * 8 + sub;
        const int q = c8 ^ ((r >> 1) & 7);
        voff[j] = (uint32_t)(r * g.lda * ESZ + q * 16);
    }
    const int dst_piece = wave * 8 * 1024;
    // BNN: lane -> row (lane%16)/4 of block lane/16 of the piece, 16-byte chunk lane%4 (gemm_lp256w4.hip); piece j adds the
    // wave-uniform (j/2) * 4 k-rows + (j%2) * 256 bytes
    const uint32_t voff_nn = BNN ? (uint32_t)(((lane & 15) >> 2) * g.ldb * ESZ + (lane >> 4) * 64 + (lane & 3) * 16) : 0u;
    const int64_t ldb4 = (int64_t)4 * g.ldb * ESZ;          // bytes between block rows (4 k-rows) of row-major B

    const int f = (l31 >> 1) & 7;
    const int rowoff_a = (wm * 128 + l31) * ROW_BYTES;
    const int rowoff_b = BNN ? wn * 4 * 256 + ((lane & 15) >> 2) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8 : (wn * 128 + l31) * ROW_BYTES;

    f32x16 acc[4][4];            // never zeroed: the first k-step of a tile accumulates into a literal zero operand
    frag fa[2][4], fb[2][4];
    tile_src iss;                // the tile whose K-tiles are being ISSUED (two ahead of the MFMAs)
    u32x4 P[3][4][2];            // the finished tile's row blocks 0..2, packed 16-bit: [i][j][p] = columns j*32 + 16p + 8h' .. of row l31

    auto read_one = [&](auto buf, auto idx, const char *pa, const char *pb) {
        constexpr int BUF = decltype(buf)::value, R = decltype(idx)::value;
        if constexpr (R >= 1 && R <= 4) fa[BUF][R - 1] = *reinterpret_cast<const frag *>(pa + (R - 1) * 32 * ROW_BYTES);
        else if constexpr (BNN) {
            constexpr int JB = (R == 0) ? 0 : R - 4;           // column block JB: k 0..3 from block row a, k 4..7 from a + 1 (2 KiB on)
            typedef short s16x4 __attribute__((ext_vector_type(4)));
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            const auto q = (__attribute__((address_space(3))) s16x4 *)(pb + JB * 256);
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(q);
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(q + 256);
            fb[BUF][JB] = __builtin_bit_cast(frag, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        } else if constexpr (R == 0) fb[BUF][0] = *reinterpret_cast<const frag *>(pb);
        else fb[BUF][R - 4] = *reinterpret_cast<const frag *>(pb + (R - 4) * 32 * ROW_BYTES);
    };
    auto dma_one = [&](auto is_b, auto jj, int64_t koff, char *base) {
        constexpr int J = decltype(jj)::value;
        if constexpr (BNN && decltype(is_b)::value)   // koff = K-tile * 128 bytes along K = K-tile * 64 k-rows of ldb elements here
            glds16_s<J * 1024>(iss.ub + koff * g.ldb + (J >> 1) * ldb4 + (J & 1) * 256, voff_nn, lds_addr_of(base));
        else
            glds16_s<J * 1024>((decltype(is_b)::value ? iss.ub : iss.ua) + koff, voff[J], lds_addr_of(base));
    };
    auto mfma_one = [&](auto buf, auto idx, auto first) {
        constexpr int BUF = decltype(buf)::value, I = decltype(idx)::value & 3, J = decltype(idx)::value >> 2;
        if constexpr (decltype(first)::value) {
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[I][J] = lp<DT>::mfma(fb[BUF][J], fa[BUF][I], zero);
        } else {
            acc[I][J] = lp<DT>::mfma(fb[BUF][J], fa[BUF][I], acc[I][J]);
        }
    };
    // hipcc picks the register half of MFMA results by a pressure heuristic; with 96 held registers beside them it has put
    // the 256 accumulators into VGPRs + scratch (849 spills).  An empty asm with an "a" constraint per block, once per
    // tile, pins them to the accumulator half (what the "a"-constraint reads of gemm_lp256p.hip do as a side effect).
    auto pin_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" : "+a"(acc[i][j]));
    };
    // block IDX = (I = IDX & 3, J = IDX >> 2) is final: pack it (I < 3; row block 3 leaves through LDS at the boundary)
    int stage_off = 0;                  // this lane's row of the boundary staging image (set before a tile's last K-tile)
    const uint32_t x16 = (uint32_t)(l31 & 15) << 4;   // chunk swizzle of the staging image: chunk ^ (row & 15)
    auto drain_one = [&](auto idx) {
        constexpr int I = decltype(idx)::value & 3, J = decltype(idx)::value >> 2;
        if constexpr (I == 3 && !Q_LDSDRAIN) {
        } else if constexpr (I == 3) {  // row block 3: packed into this wave's 8 KiB of the dead B slot (32 rows x 256 B, chunk ^ row)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u32x2 w = {lp<DT>::pack2(acc[3][J][4 * q + 0], acc[3][J][4 * q + 1]), lp<DT>::pack2(acc[3][J][4 * q + 2], acc[3][J][4 * q + 3])};
                *reinterpret_cast<u32x2 *>(smem + stage_off + (((J * 4 + q) << 4) ^ opaque(x16))) = w;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                P[I][J][q >> 1][(q & 1) * 2 + 0] = lp<DT>::pack2(acc[I][J][4 * q + 0], acc[I][J][4 * q + 1]);
                P[I][J][q >> 1][(q & 1) * 2 + 1] = lp<DT>::pack2(acc[I][J][4 * q + 2], acc[I][J][4 * q + 3]);
            }
        }
    };

    // ---- where the HELD tile goes -----------------------------------------------------------------------------------
    // Partial-line stores are poison on this memory system: the first form of this kernel stored a lane's 16 bytes on 32
    // different rows per instruction (32-byte pieces of 128-byte lines) and lost 20-35 % to the plain persistent kernel --
    // the stores crowd the path the LDS-DMA loads use.  So every store instruction must write WHOLE lines, which means a
    // transposition between "lane = row" (how the matrix core leaves the data) and "8 lanes = one 128-byte line"; with the
    // LDS full it happens in registers:
    //   1. v_permlane32_swap pairs (guide T21): a lane's two 8-byte groups q = 2p, 2p+1 of column block j become one
    //      16-byte chunk -- lanes 0-31 the even, lanes 32-63 the odd chunk of the 32-byte piece (j, p);
    //   2. a 4 x 4 transposition between the four lanes of a quad (rows 4a .. 4a+3) and the four chunks (j & 1, p) of a
    //      128-byte segment jj = j >> 1: two butterfly stages of quad-permute DPP selects (lane ^ 1 with chunk ^ 1, lane ^ 2
    //      with chunk ^ 2).  Afterwards chunk register c of lane (4a + b, h) holds row 4a + c, bytes 32 b + 16 h of the
    //      segment: one store instruction per (segment, c) writes 8 rows x 128 contiguous bytes.
    // 96 VALU operations per row block and wave, issued in the K-tiles that store the segment.
    char *hbase = nullptr;
    const uint32_t pvoff = (uint32_t)((l31 >> 2) * 4 * g.ldc * CSZ + (l31 & 3) * 32 + 16 * h);
    const int64_t rowblock = (int64_t)32 * g.ldc * CSZ, rowbytes = g.ldc * CSZ;
    const bool odd1 = (l31 & 1) != 0, odd2 = (l31 & 2) != 0;
    // chunk register c (0..3) of segment JJ of row block I lives in P[I][2 JJ + (c >> 1)][c & 1]
#define Q_CHUNK(I, JJ, c) P[I][2 * (JJ) + ((c) >> 1)][(c) & 1]
    // The transposition of one segment in 20 slices of <= 4 VALU operations (0-3: step 1 of chunk s; 4-11: step 2a, chunk
    // pair (s-4)/4*2, word (s-4)%4; 12-19: step 2b, chunk (s-12)/4, word (s-12)%4).  Slices must run in this order; they
    // are slotted between the MFMAs of k-step 2 of the K-tile that stores the segment's first row.
    auto transpose_slice = [&](auto ii, auto jjj, auto ss) {
        constexpr int I = decltype(ii)::value, JJ = decltype(jjj)::value, SL = decltype(ss)::value;
        if constexpr (SL < 4) {
            u32x4 r = Q_CHUNK(I, JJ, SL);
            const auto s0 = __builtin_amdgcn_permlane32_swap(r[0], r[2], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(r[1], r[3], false, false);
            r[0] = s0[0]; r[2] = s0[1]; r[1] = s1[0]; r[3] = s1[1];
            Q_CHUNK(I, JJ, SL) = r;
        } else if constexpr (SL < 12) {
            constexpr int c = ((SL - 4) >> 2) * 2, w = (SL - 4) & 3;                  // lane ^ 1 <-> chunk ^ 1 (quad_perm [1,0,3,2])
            u32x4 ra = Q_CHUNK(I, JJ, c), rb = Q_CHUNK(I, JJ, c + 1);
            const uint32_t a = ra[w], b = rb[w];
            const uint32_t ax = (uint32_t)__builtin_amdgcn_mov_dpp((int)a, 0xB1, 0xF, 0xF, true);
            const uint32_t bx = (uint32_t)__builtin_amdgcn_mov_dpp((int)b, 0xB1, 0xF, 0xF, true);
            ra[w] = odd1 ? bx : a;
            rb[w] = odd1 ? b : ax;
            Q_CHUNK(I, JJ, c) = ra; Q_CHUNK(I, JJ, c + 1) = rb;
        } else {
            constexpr int c = (SL - 12) >> 2, w = (SL - 12) & 3;                      // lane ^ 2 <-> chunk ^ 2 (quad_perm [2,3,0,1])
            u32x4 ra = Q_CHUNK(I, JJ, c), rb = Q_CHUNK(I, JJ, c + 2);
            const uint32_t a = ra[w], b = rb[w];
            const uint32_t ax = (uint32_t)__builtin_amdgcn_mov_dpp((int)a, 0x4E, 0xF, 0xF, true);
            const uint32_t bx = (uint32_t)__builtin_amdgcn_mov_dpp((int)b, 0x4E, 0xF, 0xF, true);
            ra[w] = odd2 ? bx : a;
            rb[w] = odd2 ? b : ax;
            Q_CHUNK(I, JJ, c) = ra; Q_CHUNK(I, JJ, c + 2) = rb;
        }
    };
    auto transpose_segment = [&](auto ii, auto jjj) {               // all of it at once (the last tile's flush)
        transpose_slice(ii, jjj, IC<0>{}); transpose_slice(ii, jjj, IC<1>{}); transpose_slice(ii, jjj, IC<2>{}); transpose_slice(ii, jjj, IC<3>{});
        transpose_slice(ii, jjj, IC<4>{}); transpose_slice(ii, jjj, IC<5>{}); transpose_slice(ii, jjj, IC<6>{}); transpose_slice(ii, jjj, IC<7>{});
        transpose_slice(ii, jjj, IC<8>{}); transpose_slice(ii, jjj, IC<9>{}); transpose_slice(ii, jjj, IC<10>{}); transpose_slice(ii, jjj, IC<11>{});
        transpose_slice(ii, jjj, IC<12>{}); transpose_slice(ii, jjj, IC<13>{}); transpose_slice(ii, jjj, IC<14>{}); transpose_slice(ii, jjj, IC<15>{});
        transpose_slice(ii, jjj, IC<16>{}); transpose_slice(ii, jjj, IC<17>{}); transpose_slice(ii, jjj, IC<18>{}); transpose_slice(ii, jjj, IC<19>{});
    };
    // the slices that follow MFMA idx of k-step 2: one behind each of MFMAs 0-7 (next to a fragment read), 2,1,2,1,.. behind 8-15
    auto transpose_gap = [&](auto tseg, auto idx) {
        constexpr int TS = decltype(tseg)::value, IDX = decltype(idx)::value;
        if constexpr (TS >= 0) {
            constexpr int first = IDX < 8 ? IDX : 8 + ((IDX - 8) * 3 + 1) / 2, count = IDX < 8 ? 1 : 2 - ((IDX - 8) & 1);   // 8,10,11,13,14,16,17,19
            if constexpr (TS < 2 || TS == 2) {
                constexpr int SEG = TS == 2 ? 0 : TS;
                transpose_slice(IC<0>{}, IC<SEG>{}, IC<first>{});
                if constexpr (count == 2) transpose_slice(IC<0>{}, IC<SEG>{}, IC<(count == 2 ? first + 1 : first)>{});
            }
            if constexpr (TS == 2) {                                    // D == 8: both segments in this K-tile
                transpose_slice(IC<0>{}, IC<1>{}, IC<first>{});
                if constexpr (count == 2) transpose_slice(IC<0>{}, IC<1>{}, IC<(count == 2 ? first + 1 : first)>{});
            }
        }
    };
    // store N (0..7) of held row block I: segment JJ = N / 4 (bytes 128 JJ .. of the wave's 256-byte rows), chunk register
    // c = N % 4 = row 4a + c of every quad a.  `rb_base` = wave-uniform address of the row block's first row.
    auto store_one = [&](auto ii, auto nn, char *rb_base) {
        constexpr int I = decltype(ii)::value, N = decltype(nn)::value, JJ = N / 4, CC = N % 4;
        if constexpr (!Q_SLICED && CC == 0) transpose_segment(IC<I>{}, IC<JJ>{});
        const u32x4 v = Q_CHUNK(I, JJ, CC);
        if (Q_ABL & 1) { asm volatile("" ::"v"(v)); return; }
#if Q_DUMMY
        rb_base = static_cast<char *>(g.c) + wave * 8192;
#endif
#if Q_STORE_FORM == 1
        const uint32_t so_ = Q_DUMMY ? (uint32_t)(lane * 16) : pvoff;
        char *sb_ = rb_base + (Q_DUMMY ? 0 : CC * rowbytes);
        asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3 nt\n\ts_nop 1" ::"v"(so_), "v"(v), "s"(sb_), "n"(JJ * 128) : "memory");
#else
        char *dst = Q_DUMMY ? rb_base + lane * 16 + JJ * 1024 : rb_base + CC * rowbytes + JJ * 128 + pvoff;
#if Q_NT
        __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(dst));
#else
        *reinterpret_cast<u32x4 *>(dst) = v;
#endif
#endif
    };
    // the D stores of group G (static) of the row block in P[0]
    auto store_group = [&](auto gg, char *rb_base) {
        constexpr int G = decltype(gg)::value;
        store_one(IC<0>{}, IC<G * D + 0>{}, rb_base);
        if constexpr (D >= 2) store_one(IC<0>{}, IC<G * D + (D >= 2 ? 1 : 0)>{}, rb_base);
        if constexpr (D >= 4) { store_one(IC<0>{}, IC<G * D + (D >= 4 ? 2 : 0)>{}, rb_base); store_one(IC<0>{}, IC<G * D + (D >= 4 ? 3 : 0)>{}, rb_base); }
        if constexpr (D >= 8) { store_one(IC<0>{}, IC<G * D + (D >= 8 ? 4 : 0)>{}, rb_base); store_one(IC<0>{}, IC<G * D + (D >= 8 ? 5 : 0)>{}, rb_base);
                                store_one(IC<0>{}, IC<G * D + (D >= 8 ? 6 : 0)>{}, rb_base); store_one(IC<0>{}, IC<G * D + (D >= 8 ? 7 : 0)>{}, rb_base); }
    };

    // One k-step, instruction order pinned by hand: MFMA idx, then at most one fragment read of the NEXT k-step or one DMA
    // piece; FIRST: the MFMAs take a zero C operand; DRAIN: block idx - 2 is packed behind MFMA idx (its own MFMA retired
    // two matrix-pipe slots ago, so the accumulator read meets no pipeline hazard and the pipe never waits for the VALU).
#define Q_STEP_BODY(CUR, NXT, RMASK, DMASK, IS_B, J0, FIRST, DRAIN, TSEG)                             \
    {                                                                                                \
        constexpr unsigned rmask_ = (RMASK), dmask_ = (DMASK);                                       \
        Q_GROUP(CUR, NXT, 0, IS_B, J0, FIRST, DRAIN, TSEG)  Q_GROUP(CUR, NXT, 1, IS_B, J0, FIRST, DRAIN, TSEG)   \
        Q_GROUP(CUR, NXT, 2, IS_B, J0, FIRST, DRAIN, TSEG)  Q_GROUP(CUR, NXT, 3, IS_B, J0, FIRST, DRAIN, TSEG)   \
        Q_GROUP(CUR, NXT, 4, IS_B, J0, FIRST, DRAIN, TSEG)  Q_GROUP(CUR, NXT, 5, IS_B, J0, FIRST, DRAIN, TSEG)   \
        Q_GROUP(CUR, NXT, 6, IS_B, J0, FIRST, DRAIN, TSEG)  Q_GROUP(CUR, NXT, 7, IS_B, J0, FIRST, DRAIN, TSEG)   \
        Q_GROUP(CUR, NXT, 8, IS_B, J0, FIRST, DRAIN, TSEG)  Q_GROUP(CUR, NXT, 9, IS_B, J0, FIRST, DRAIN, TSEG)   \
        Q_GROUP(CUR, NXT, 10, IS_B, J0, FIRST, DRAIN, TSEG) Q_GROUP(CUR, NXT, 11, IS_B, J0, FIRST, DRAIN, TSEG)  \
        Q_GROUP(CUR, NXT, 12, IS_B, J0, FIRST, DRAIN, TSEG) Q_GROUP(CUR, NXT, 13, IS_B, J0, FIRST, DRAIN, TSEG)  \
        Q_GROUP(CUR, NXT, 14, IS_B, J0, FIRST, DRAIN, TSEG) Q_GROUP(CUR, NXT, 15, IS_B, J0, FIRST, DRAIN, TSEG)  \
    }
#define Q_GROUP(CUR, NXT, IDX, IS_B, J0, FIRST, DRAIN, TSEG)                                         \
    mfma_one(IC<CUR>{}, IC<IDX>{}, IC<FIRST>{});                                                     \
    if constexpr ((rmask_ >> IDX) & 1u)                                                              \
        read_one(IC<NXT>{}, IC<__builtin_popcount(rmask_ & ((1u << IDX) - 1u))>{}, rd_a, rd_b);      \
    if constexpr ((dmask_ >> IDX) & 1u)                                                              \
        dma_one(IC<IS_B>{}, IC<J0 + __builtin_popcount(dmask_ & ((1u << IDX) - 1u))>{}, dma_koff, dma_base); \
    if constexpr ((DRAIN) && (IDX) >= 2) {        /* MFMA first: the pack VALU runs under it, not in front of it */  \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        drain_one(IC<((IDX) >= 2 ? (IDX) - 2 : 0)>{});                                               \
    }                                                                                                \
    if constexpr ((TSEG) >= 0) {                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        transpose_gap(IC<(TSEG)>{}, IC<IDX>{});                                                      \
    }                                                                                                \
    __builtin_amdgcn_sched_barrier(0);

    // ---- first tile of this workgroup: units 0..3 (its K-tiles 0 and 1), then the first fragments ---------
    uint32_t L = blockIdx.x;
    tile_src cur = locate(L);
    iss = cur;
    {
        const int64_t k0 = 0, k1 = ROW_BYTES;
        char *b0 = smem + dst_piece;
#define Q_PRO(IS_B, KOFF, SLOT)                                                                      \
        dma_one(IC<IS_B>{}, IC<0>{}, KOFF, b0 + SLOT * UNIT_BYTES); dma_one(IC<IS_B>{}, IC<1>{}, KOFF, b0 + SLOT * UNIT_BYTES); \
        dma_one(IC<IS_B>{}, IC<2>{}, KOFF, b0 + SLOT * UNIT_BYTES); dma_one(IC<IS_B>{}, IC<3>{}, KOFF, b0 + SLOT * UNIT_BYTES); \
        dma_one(IC<IS_B>{}, IC<4>{}, KOFF, b0 + SLOT * UNIT_BYTES); dma_one(IC<IS_B>{}, IC<5>{}, KOFF, b0 + SLOT * UNIT_BYTES); \
        dma_one(IC<IS_B>{}, IC<6>{}, KOFF, b0 + SLOT * UNIT_BYTES); dma_one(IC<IS_B>{}, IC<7>{}, KOFF, b0 + SLOT * UNIT_BYTES);
        Q_PRO(0, k0, 0) Q_PRO(1, k0, 1) Q_PRO(0, k1, 2) Q_PRO(1, k1, 3)
#undef Q_PRO
    }
    WAIT_VMCNT(16);                      // units 0, 1 landed (this wave's share)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    {
        const int x = (h ^ f) << 4;
        const char *rd_a = smem + rowoff_a + x, *rd_b = smem + UNIT_BYTES + rowoff_b + (BNN ? h * 4096 : x);
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
    // B fragment offsets per k-step: the same chunks as A for [N][K]; block rows a = 4s + 2h (8 blocks of 256 B each) for row-major B
    const int y0 = BNN ? h * 4096 : x0, y1 = BNN ? 8192 + h * 4096 : x1, y2 = BNN ? 16384 + h * 4096 : x2, y3 = BNN ? 24576 + h * 4096 : x3;

    char *__restrict__ C = static_cast<char *>(g.c);
    uint32_t Lnext = 0;
    bool has_next = false;
    tile_src nxt = cur;
    int kbase = 0;                       // K-tile index of the issue side = t + 2 - kbase
    int t = 0;

    // Every K-tile ends its basic block with a never-taken branch (two scalar instructions).  hipcc's instruction selection
    // schedules per basic block for register pressure, and MFMAs carry no ordering edge: with eight K-tile bodies in one
    // block next to 96 held registers it sank all 512 MFMAs to the end of the block and spilled every fragment on the way
    // (849 spills).  One block per K-tile keeps each MFMA where the schedule puts it.
#define Q_BLOCK_END() if (__builtin_expect(t > 0x3fffffff, 0)) asm volatile("s_trap 2");
    // One K-tile (schedule of gemm_lp256w4.hip).  FIRST: K-tile 0 of an output tile (zero C operand in k-step 0).
    // LASTK: the tile's last K-tile (blocks are packed as their last MFMA retires).  WAITN: LDS-DMA pieces + stores that
    // may still be in flight at the hand-over (exactly what was issued after unit 2t+3); WAITN0 instead when `w0` is set
    // (the first K-tile of the drip phase has no dripped stores in front of it).  G >= 0: store group G of the row block
    // in P[0] behind the last DMA piece.
#define Q_KTILE(FIRST, LASTK, WAITN, WAITN0, G)                                                              \
    {                                                                                                       \
        const int sa1 = adv(sa, 2), sb1 = adv(sb, 2);     /* units of K-tile t+1 */                          \
        const int s4 = adv(sa, 4);                        /* unit 2t+4 -> slot of unit 2t-1 */               \
        const int s5 = sa;                                /* unit 2t+5 -> slot of unit 2t   */               \
        if (t == nk - 2 && has_next) { iss = nxt; kbase = nk; }   /* from here on the stream feeds the next tile */ \
        const int64_t dma_koff = (int64_t)min(t + 2 - kbase, nk - 1) * ROW_BYTES;   /* clamp: only without a next tile */ \
        const char *rd_a, *rd_b;                                                                            \
        char *dma_base;                                                                                     \
        rd_a = smem + sa + rowoff_a + x1; rd_b = smem + sb + rowoff_b + y1; dma_base = smem + s4 + dst_piece; \
        /* the segment whose first row this K-tile stores is transposed between the MFMAs of k-step 2 */  \
        constexpr int n0_ = ((G) >= 0 ? (G) : 0) * D;                                                       \
        constexpr int tseg_ = (Q_SLICED && (G) >= 0 && n0_ % 4 == 0) ? (D == 8 ? 2 : n0_ / 4) : -1;         \
        Q_STEP_BODY(0, 1, 0x00FFu, 0xAA00u, 0, 0, FIRST, 0, -1)                                              \
        rd_a = smem + sa + rowoff_a + x2; rd_b = smem + sb + rowoff_b + y2;                                 \
        Q_STEP_BODY(1, 0, 0x00FFu, 0xAA00u, 0, 4, 0, 0, -1)                                                  \
        rd_a = smem + sa + rowoff_a + x3; rd_b = smem + sb + rowoff_b + y3;                                 \
        Q_STEP_BODY(0, 1, 0x00FFu, 0x0000u, 0, 0, 0, 0, tseg_)                                               \
        if constexpr (Q_STORE_AT == 2 && (G) >= 0) { store_group(IC<((G) >= 0 ? (G) : 0)>{}, rb_base); __builtin_amdgcn_sched_barrier(0); } \
        if constexpr (Q_STORE_AT == 2) { if constexpr ((G) >= 0) WAIT_VMCNT(8 + D); else WAIT_VMCNT(((FIRST) && (WAITN) == 16) ? 16 : 8); }   \
        else if constexpr ((WAITN0) != (WAITN)) { if (w0) WAIT_VMCNT(WAITN0); else WAIT_VMCNT(WAITN); }     \
        else WAIT_VMCNT(WAITN);          /* my share of the next K-tile landed */                            \
        WAIT_LGKM0();                    /* my reads of this K-tile are complete */                          \
        __builtin_amdgcn_s_barrier();    /* BAR_t */                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        rd_a = smem + sa1 + rowoff_a + x0; rd_b = smem + sb1 + rowoff_b + y0; dma_base = smem + s5 + dst_piece; \
        if constexpr (LASTK) stage_off = sb + wave * 8192 + (int)opaque((uint32_t)(l31 * 256 + 8 * h));   /* B unit of this K-tile: dead since BAR_t */ \
        Q_STEP_BODY(1, 0, 0x5555u, 0xAAAAu, 1, 0, 0, LASTK, -1)                                              \
        if constexpr (LASTK) { drain_one(IC<14>{}); drain_one(IC<15>{}); __builtin_amdgcn_sched_barrier(0); } \
        if constexpr (Q_STORE_AT == 3 && (G) >= 0) { store_group(IC<((G) >= 0 ? (G) : 0)>{}, rb_base); __builtin_amdgcn_sched_barrier(0); } \
        if constexpr (FIRST) pin_acc();  /* (no instruction: keeps the accumulators in the AGPR half) */     \
        sa = sa1;                                                                                           \
        sb = sb1;                                                                                           \
        ++t;                                                                                                \
        Q_BLOCK_END()                                                                                       \
    }

    bool held = false;                   // P holds a finished tile whose stores are still to be issued
#ifdef Q_TRACE
    int qt_tile = 0;
#endif
    constexpr int GPR = 8 / D;           // K-tiles (store groups) per held row block
    for (;;) {
        Lnext = L + gridDim.x;
        has_next = Lnext < total;
        nxt = cur;
        if (has_next) nxt = locate(Lnext);
        kbase = 0;
        t = 0;
        bool w0 = false;
        char *rb_base = hbase;
        Q_STAMP(0);
        if (!held) {
            Q_KTILE(1, 0, 8, 8, -1)
#pragma nounroll                          /* (hipcc once unrolled this loop 8 x and spilled every fragment read) */
            while (t < nk - 1) Q_KTILE(0, 0, 8, 8, -1)
        } else {
            // K-tile 0: the 8 boundary stores of the previous tile sit between unit 3 and unit 4 of this stream
            Q_KTILE(1, 0, 16, 16, -1)
            // drip phase, K-tiles 1 .. 24 / D: the row block in P[0] leaves, D stores behind every K-tile's last DMA piece,
            // then the next row block moves down (64 register moves per 8 / D K-tiles).  Its first K-tile has no dripped
            // stores in front of it (w0), every other one has the D of its predecessor.
            w0 = true;
#pragma nounroll
            for (int rb = 0; rb < 3; ++rb) {
                constexpr int WD = 8 + D + Q_WAIT_EXTRA;
                Q_KTILE(0, 0, WD, 8, 0)
                w0 = false;
                if constexpr (GPR > 1) Q_KTILE(0, 0, WD, WD, 1)
                if constexpr (GPR > 2) { Q_KTILE(0, 0, WD, WD, 2) Q_KTILE(0, 0, WD, WD, 3) }
                if constexpr (GPR > 4) { Q_KTILE(0, 0, WD, WD, 4) Q_KTILE(0, 0, WD, WD, 5) Q_KTILE(0, 0, WD, WD, 6) Q_KTILE(0, 0, WD, WD, 7) }
#pragma unroll
                for (int j = 0; j < 4; ++j) { P[0][j][0] = P[1][j][0]; P[0][j][1] = P[1][j][1]; P[1][j][0] = P[2][j][0]; P[1][j][1] = P[2][j][1]; }
                rb_base += rowblock;
            }
            Q_KTILE(0, 0, 8 + D + Q_WAIT_EXTRA, 8 + D + Q_WAIT_EXTRA, -1)   // the last dripped stores are still allowed in flight here
#pragma nounroll
            while (t < nk - 1) Q_KTILE(0, 0, 8, 8, -1)
        }
        Q_KTILE(0, 1, 8, 8, -1)                                   // t == nk - 1: row blocks 0..2 are packed into P
        Q_STAMP(1);

        // ---- tile boundary: row block 3 through this wave's 8 KiB of the dead B slot (as gemm_lp256p.hip) ----------
        {
            char *stage = smem + adv(sb, 3) + wave * 8192;          // slot of the last B unit, my DMA region of it
            const int64_t cbase = cur.batch * g.stride_c;
            char *wbase = C + (cbase + (cur.m0 + wm * 128) * g.ldc + cur.n0 + wn * 128) * CSZ;   // my 128x128 block
            char *crow = wbase + (int64_t)(96 + lane / 16) * g.ldc * CSZ + (lane % 16) * 16;
            const int64_t cstep = (int64_t)4 * g.ldc * CSZ;
            // (row block 3 was packed into the staging image as its blocks retired, during the last k-step)
            if constexpr (!Q_LDSDRAIN) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        char *d = stage + opaque((uint32_t)(l31 * 256 + 8 * h)) + (((j * 4 + q) << 4) ^ opaque(x16));
                        u32x2 w = {lp<DT>::pack2(acc[3][j][4 * q + 0], acc[3][j][4 * q + 1]), lp<DT>::pack2(acc[3][j][4 * q + 2], acc[3][j][4 * q + 3])};
                        *reinterpret_cast<u32x2 *>(d) = w;
                        if ((q & 1) == 1) __builtin_amdgcn_sched_barrier(0);
                    }
            }
            WAIT_LGKM0();                                          // same-wave hand-over: DS ops of one wave execute in order
            // row r = 4 it + lane / 16, chunk (lane % 16) ^ (r & 15): the lane part of the address is one value, the `it` part
            // an XOR of bits 6-7 and an immediate offset
            const uint32_t rdx = (uint32_t)((lane >> 4) * 256 + (((lane & 15) ^ (lane >> 4)) << 4));
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const u32x4 v = *reinterpret_cast<const u32x4 *>(stage + it * 1024 + (opaque(rdx) ^ (uint32_t)((it & 3) << 6)));
                if (!(Q_ABL & 2)) __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(crow + it * cstep));
                else asm volatile("" ::"v"(v));
            }
            WAIT_LGKM0();                                          // staged rows are in registers before this wave's next DMA lands there
            __builtin_amdgcn_sched_barrier(0);
            hbase = wbase;
            held = true;
        }
        Q_STAMP(2);
#ifdef Q_TRACE
        ++qt_tile;
#endif
        if (!has_next) break;
        cur = nxt;
        L = Lnext;
    }
#undef Q_KTILE
#undef Q_STEP_BODY
#undef Q_GROUP
    // ---- the last tile of this workgroup has no K loop to hide under: its held stores leave at once ------------------
#define Q_FLUSH(I)                                                                                       \
    if constexpr (Q_SLICED) { transpose_segment(IC<I>{}, IC<0>{}); transpose_segment(IC<I>{}, IC<1>{}); }      \
    store_one(IC<I>{}, IC<0>{}, hbase + (I) * rowblock); store_one(IC<I>{}, IC<1>{}, hbase + (I) * rowblock); \
    store_one(IC<I>{}, IC<2>{}, hbase + (I) * rowblock); store_one(IC<I>{}, IC<3>{}, hbase + (I) * rowblock); \
    store_one(IC<I>{}, IC<4>{}, hbase + (I) * rowblock); store_one(IC<I>{}, IC<5>{}, hbase + (I) * rowblock); \
    store_one(IC<I>{}, IC<6>{}, hbase + (I) * rowblock); store_one(IC<I>{}, IC<7>{}, hbase + (I) * rowblock);
    Q_FLUSH(0) Q_FLUSH(1) Q_FLUSH(2)
#undef Q_FLUSH
    WAIT_VMCNT(0);                       // drain the clamped tail DMA (and the last stores) before the workgroup retires
}

template <int DT, int D, bool BNN>
void launch(mi355_ctx *ctx, hipStream_t s, const gemm_args &g, uint32_t batch)
{
    lds_opt_in(ctx, reinterpret_cast<const void *>(gemm_lp256q_kernel<DT, D, BNN>), LDS_BYTES);
    const uint32_t total = g.tiles_m * g.tiles_n * batch;
    const uint32_t grid = std::min<uint32_t>(total, ctx->props.num_streaming_multiprocessors);   // one workgroup per CU (LDS admits no more)
    hipLaunchKernelGGL((gemm_lp256q_kernel<DT, D, BNN>), dim3(grid), dim3(256), LDS_BYTES, s, g);
}

int drip_for(int64_t nk)                 // fewest stores per K-tile whose drip phase (K-tiles 1 .. 24 / D) fits: nk >= 24 / D + 3
{
    if (Q_FORCE_D) return nk >= 24 / (Q_FORCE_D ? Q_FORCE_D : 1) + 3 ? Q_FORCE_D : 0;
    return nk >= 27 ? 1 : nk >= 15 ? 2 : nk >= 9 ? 4 : nk >= 6 ? 8 : 0;
}

}  // namespace

#ifdef Q_TRACE
extern "C" __attribute__((visibility("default"))) int mi355_dev_q_trace(unsigned long long *host_out)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(q_trace_buf), sizeof(unsigned long long) * 256 * 32);
}
#endif

namespace mi355 {

bool gemm_lp256q_supports(const mi355_gemm_desc &d, const void *a, const void *b, const void *c)
{
    if (d.dtype_ab != MI355_DTYPE_BF16 && d.dtype_ab != MI355_DTYPE_F16) return false;
    if (d.dtype_c != d.dtype_ab) return false;                     // (f32 C would need 256 held registers: gemm_lp256p.hip)
    if (!gemm_lp256p_supports(d, a, b, c)) return false;           // full tiles, K-contiguous 16-byte aligned operands, ...
    if (drip_for(d.k / 64) == 0) return false;
    if (d.trans_b && d.lda != d.ldb) return false;                 // [N][K] B: one set of per-lane DMA offsets for both operands
    if ((int64_t)32 * d.ldc * 2 >= (1ll << 32)) return false;      // per-lane store offsets are 32-bit
    return true;
}

int32_t launch_gemm_lp256q(mi355_ctx *ctx, hipStream_t s, const mi355_gemm_desc &d, const void *a, const void *b, void *c)
{
    if (!gemm_lp256q_supports(d, a, b, c))
        return fail(ctx, MI355_E_UNSUPPORTED, "lp256q GEMM: shape/layout not supported by this kernel");
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
#define Q_LAUNCH(DD)                                                                              \
    if (d.trans_b) { if (bf) launch<MI355_DTYPE_BF16, DD, false>(ctx, s, g, batch); else launch<MI355_DTYPE_F16, DD, false>(ctx, s, g, batch); } \
    else { if (bf) launch<MI355_DTYPE_BF16, DD, true>(ctx, s, g, batch); else launch<MI355_DTYPE_F16, DD, true>(ctx, s, g, batch); }
    if (drip == 1) { Q_LAUNCH(1) } else if (drip == 2) { Q_LAUNCH(2) } else if (drip == 4) { Q_LAUNCH(4) } else { Q_LAUNCH(8) }
#undef Q_LAUNCH
    check_launch(ctx, "mi355_gemm(lp256q)");
    return MI355_OK;
}

}  // namespace mi355
