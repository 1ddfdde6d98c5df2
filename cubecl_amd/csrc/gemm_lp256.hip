// gemm_lp256.hip -- bf16 / f16 GEMM, 256x256x64 workgroup tile, 8 waves in two "ping-pong" groups.
//
// Roofline: MFMA bf16/f16, ~2.5 PFLOP/s dense (MI355X_MICROARCH.md).  Why this shape: a 128x128
// tile needs 64 B/clk/CU of L2->LDS traffic at the MFMA rate (more than the LDS-DMA path delivers);
// 256x256x64 needs 32 B/clk/CU.
//
// Geometry
//   8 waves = 2 (M) x 4 (N); wave (g, wc) owns a 128 x 64 output = 4 x 2 MFMA tiles of 32x32
//   (v_mfma_f32_32x32x16, 128 accumulator registers).  g = wave >> 2 is also the ping-pong group:
//   waves w and w+4 share a SIMD, one from each group.
//   LDS (all 160 KiB of the CU): A ring of 2 x 32 KiB + B ring of 3 x 32 KiB, rows of one
//   128-byte line (64 k-values), filled by LDS-DMA (global_load_lds_dwordx4), XOR-swizzled on the
//   SOURCE address and on the fragment read exactly as in gemm_lp128.hip (conflict-free
//   ds_read_b128).
//
// Schedule (per K-tile t, per wave): two phases, one per 32-deep k-half, each
//       {L: 12 fragment reads + 4 DMA issues}  barrier  {C: 16 MFMAs}  barrier.
//   Group 1 runs one barrier behind group 0, so on every SIMD one wave is in its MFMA segment while
//   its partner reads LDS and issues DMA (guide section 5, T3+T4+T5).  Four barriers per K-tile
//   (the first version of this kernel split the tile into four row quadrants = eight barriers; the
//   ablation in profiles/ showed ~200 cycles of fixed cost per barrier interval, so fewer and longer
//   phases win).
//       L(t,0): read k 0-31 of A(t), B(t)      issue A[g](t+1) -> A ring slot (t+1)%2
//       L(t,1): read k 32-63                   issue B(t+2)    -> B ring slot (t+2)%3
//   The A half of a group is private to it (loaded and read by the same 4 waves); B is loaded by all
//   8 waves and read by both groups, which is why B gets the third ring slot: every DMA is issued
//   >= 1 interval after the last read of the slot it overwrites and >= 3 intervals (~1700+ cycles)
//   before its first reader.  vmcnt never reaches 0 in the loop; the waits are counted against the
//   fixed issue order  ... A(t+1) B(t+2) A(t+2) B(t+3) ...  (4 DMA instructions per entry per wave):
//       end of L(t,1): vmcnt(8)  -> B(t+1) landed  (A(t+1), B(t+2) may still fly)
//       end of C(t,1): vmcnt(4)  -> A(t+1) landed
//   Past the last K-tile the same instructions are issued against a clamped tile index (harmless
//   re-reads into dead slots) so the counts stay uniform; the kernel drains them before its epilogue.
#include "gemm_common.hpp"

using namespace mi355;

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int ROW_BYTES = BK * 2;                 // 128
constexpr int OPER_BYTES = BM * ROW_BYTES;        // 32 KiB: one ring slot
constexpr int A_RING = 2, B_RING = 3;
constexpr int B_BASE = A_RING * OPER_BYTES;       // 64 KiB
constexpr int LDS_BYTES = (A_RING + B_RING) * OPER_BYTES;   // 160 KiB

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

__device__ __forceinline__ void glds16(const void *gsrc, void *lds_dst)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)lds_dst, 16, 0, 0);
}

// Development ablation switches (always 0 in the shipped library; tools/dev/build_variants.sh builds
// side copies): 1 = no DMA, 2 = no fragment reads, 4 = no MFMA, 8 = lgkmcnt wait after the barrier.
#ifndef LP256_ABL
#define LP256_ABL 0
#endif

// Development timing trace (-DLP256_TRACE): per-wave cycle totals of the L / barrier / C / barrier
// segments, read back with mi355_dev_lp256_trace (dev builds only).
#ifdef LP256_TRACE
__device__ unsigned long long lp256_trace_buf[64 * 8 * 12];
#define TR_DECL unsigned long long tr_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long tr_t = __builtin_amdgcn_s_memtime(); const unsigned long long tr_loop0 = tr_t;
#define TR(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tr_acc[k] += now_ - tr_t; tr_t = now_; } while (0)
#else
#define TR_DECL
#define TR(k)
#endif

#define WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define WAIT_LGKM0_L() do { if (!(LP256_ABL & 8)) WAIT_LGKM0(); } while (0)
#define PHASE_BARRIER()                      \
    do {                                     \
        __builtin_amdgcn_sched_barrier(0);   \
        __builtin_amdgcn_s_barrier();        \
        __builtin_amdgcn_sched_barrier(0);   \
    } while (0)

template <int DT, int DT_C>
__global__ void __launch_bounds__(512, 2)
gemm_lp256_kernel(gemm_args g)
{
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    typedef typename lp<DT>::frag frag;

#ifdef LP256_TRACE
    const unsigned long long tr_k0 = __builtin_amdgcn_s_memtime();
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;     // M half == ping-pong group
    const int wc = wave & 3;
    const int h = lane >> 5, l31 = lane & 31;

    uint32_t tm, tn, batch_u;
    batched_tile_coords(g.tiles_m, g.tiles_n, g.group_m, tm, tn, batch_u);
    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
    const int64_t batch = batch_u;
    const char *__restrict__ A = static_cast<const char *>(g.a) + batch * g.stride_a * 2;
    const char *__restrict__ B = static_cast<const char *>(g.b) + batch * g.stride_b * 2;
    const int nk = (int)(g.k / BK);

    // ---- DMA map: 1 KiB pieces of 8 rows; lane -> (row = piece*8 + lane/8, physical chunk = lane%8)
    //   A: my group's 128 rows = 16 pieces, this wave fills pieces wc*4 + j
    //   B: 256 rows = 32 pieces, this wave fills pieces wave*4 + j
    const char *src_a[4], *src_b[4];
    int dst_a[4], dst_b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int sub = lane >> 3, c = lane & 7;
        const int ra_ = grp * 128 + (wc * 4 + j) * 8 + sub;
        src_a[j] = A + (min(m0 + ra_, g.m - 1) * g.lda + (c ^ ((ra_ >> 1) & 7)) * 8) * 2;
        dst_a[j] = (grp * 128 + (wc * 4 + j) * 8) * ROW_BYTES;
        const int rb_ = (wave * 4 + j) * 8 + sub;
        src_b[j] = B + (min(n0 + rb_, g.n - 1) * g.ldb + (c ^ ((rb_ >> 1) & 7)) * 8) * 2;
        dst_b[j] = (wave * 4 + j) * 8 * ROW_BYTES;
    }
    auto issue = [&](const char *const (&src)[4], const int (&dst)[4], int tile, int slot_base) {
        const int64_t koff = (int64_t)min(tile, nk - 1) * (BK * 2);
        char *base = smem + slot_base;
        if (LP256_ABL & 1) return;
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(src[j] + koff, base + dst[j]);
    };

    // ---- fragment read offsets (bytes within a ring slot) -------------------------------------------
    int ra[4], fa[4], rb[2], fb[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = grp * 128 + i * 32 + l31;
        ra[i] = row * ROW_BYTES; fa[i] = (row >> 1) & 7;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = wc * 64 + j * 32 + l31;
        rb[j] = row * ROW_BYTES; fb[j] = (row >> 1) & 7;
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    frag af[4][2], bf[2][2];

    auto read_half = [&](int a_base, int b_base, int s) {
        if (LP256_ABL & 2) return;
        const char *pa = smem + a_base;
        const char *pb = smem + b_base;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int q = (2 * s + kk) * 2 + h;
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j][kk] = *reinterpret_cast<const frag *>(pb + rb[j] + ((q ^ fb[j]) << 4));
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i][kk] = *reinterpret_cast<const frag *>(pa + ra[i] + ((q ^ fa[i]) << 4));
        }
    };
    auto compute = [&]() {
        if (LP256_ABL & 8) WAIT_LGKM0();
        if (LP256_ABL & 4) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(bf[j][kk]));
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(af[i][kk]));
            }
            return;
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = lp<DT>::mfma(bf[j][kk], af[i][kk], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- prologue: B(0) A(0) B(1) in the steady-state issue order -------------------------------------
    issue(src_b, dst_b, 0, B_BASE + 0 * OPER_BYTES);
    issue(src_a, dst_a, 0, 0);
    issue(src_b, dst_b, 1, B_BASE + 1 * OPER_BYTES);
    WAIT_VMCNT(4);                       // B(0), A(0) landed (this wave's share); B(1) may fly
    PHASE_BARRIER();
    if (grp == 1) PHASE_BARRIER();       // group 1 runs one barrier behind

    TR_DECL
    int a_cur = 0;                       // byte offset of A ring slot t % 2
    int b_cur = B_BASE;                  // byte offset of B ring slot t % 3
    for (int t = 0; t < nk; ++t) {
        const int a_nxt = a_cur ^ OPER_BYTES;
        int b_nn = b_cur + 2 * OPER_BYTES;                 // slot (t + 2) % 3
        if (b_nn >= B_BASE + B_RING * OPER_BYTES) b_nn -= B_RING * OPER_BYTES;
        // ---- k-half 0 ----
        read_half(a_cur, b_cur, 0);
        issue(src_a, dst_a, t + 1, a_nxt);
        WAIT_LGKM0_L();
        TR(0);
        PHASE_BARRIER();
        TR(1);
        compute();
        TR(2);
        PHASE_BARRIER();
        TR(3);
        // ---- k-half 1 ----
        read_half(a_cur, b_cur, 1);
        issue(src_b, dst_b, t + 2, b_nn);
        WAIT_VMCNT(8);                   // B(t+1) landed; {A(t+1), B(t+2)} may fly
        WAIT_LGKM0_L();
        TR(4);
        PHASE_BARRIER();
        TR(5);
        compute();
        WAIT_VMCNT(4);                   // A(t+1) landed; B(t+2) may fly
        TR(6);
        PHASE_BARRIER();
        TR(7);
        a_cur = a_nxt;
        b_cur += OPER_BYTES;
        if (b_cur >= B_BASE + B_RING * OPER_BYTES) b_cur = B_BASE;
    }
    if (grp == 0) PHASE_BARRIER();       // re-align the barrier count of the two groups
    WAIT_VMCNT(0);                       // drain the clamped tail DMA before the workgroup retires
#ifdef LP256_TRACE
    const unsigned long long tr_loop1 = __builtin_amdgcn_s_memtime();
#endif

    // ---- epilogue: lane owns C[m][n .. n+3] per register quad -------------------------------------------
    char *__restrict__ C = static_cast<char *>(g.c);
    constexpr int CSZ = (DT_C == MI355_DTYPE_F32) ? 4 : 2;
    const int64_t cbase = batch * g.stride_c;
    const bool vec_ok = (((g.ldc * CSZ) & (4 * CSZ - 1)) == 0) &&
                        (((reinterpret_cast<uintptr_t>(C) + (uint64_t)cbase * CSZ) & (4 * CSZ - 1)) == 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t m = m0 + grp * 128 + i * 32 + l31;
        if (m >= g.m) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t n = n0 + wc * 64 + j * 32 + 8 * q + 4 * h;
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
#ifdef LP256_TRACE
    {
        const unsigned long long tr_k1 = __builtin_amdgcn_s_memtime();
        if (blockIdx.x < 64 && blockIdx.y == 0 && lane == 0) {
            unsigned long long *o = lp256_trace_buf + (blockIdx.x * 8 + wave) * 12;
            for (int k = 0; k < 8; ++k) o[k] = tr_acc[k];
            o[8] = tr_loop0 - tr_k0; o[9] = tr_loop1 - tr_loop0; o[10] = tr_k1 - tr_loop1; o[11] = tr_k1 - tr_k0;
        }
    }
#endif
}

template <int DT, int DT_C>
void launch(mi355_ctx *ctx, hipStream_t s, const gemm_args &g, uint32_t batch)
{
    lds_opt_in(ctx, reinterpret_cast<const void *>(gemm_lp256_kernel<DT, DT_C>), LDS_BYTES);
    hipLaunchKernelGGL((gemm_lp256_kernel<DT, DT_C>), dim3(g.tiles_m * g.tiles_n, batch), dim3(512), LDS_BYTES, s, g);
}

}  // namespace

#ifdef LP256_TRACE
extern "C" __attribute__((visibility("default"))) int mi355_dev_lp256_trace(unsigned long long *host_out)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(lp256_trace_buf), sizeof(unsigned long long) * 64 * 8 * 12);
}
#endif

namespace mi355 {

bool gemm_lp256_supports(const mi355_gemm_desc &d, const void *a, const void *b, const void *c)
{
    (void)c;
    if (d.dtype_ab != MI355_DTYPE_BF16 && d.dtype_ab != MI355_DTYPE_F16) return false;
    if (d.dtype_c != MI355_DTYPE_F32 && d.dtype_c != d.dtype_ab) return false;
    if (d.trans_a || !d.trans_b) return false;
    if (d.k < BK || d.k % BK != 0) return false;
    if (d.m < 1 || d.n < 1) return false;
    if ((d.lda & 7) || (d.ldb & 7) || (d.stride_a & 7) || (d.stride_b & 7)) return false;
    if ((reinterpret_cast<uintptr_t>(a) & 15u) || (reinterpret_cast<uintptr_t>(b) & 15u)) return false;
    if (d.batch > 65535) return false;
    const int64_t tiles = ((d.m + BM - 1) / BM) * ((d.n + BN - 1) / BN);
    if (tiles * std::max<int64_t>(d.batch, 1) > 0x7FFFFFFF) return false;   // 32-bit (batch, tile) sequence for the XCD remap
    return true;
}

int32_t launch_gemm_lp256(mi355_ctx *ctx, hipStream_t s, const mi355_gemm_desc &d, const void *a, const void *b,
                          void *c)
{
    if (!gemm_lp256_supports(d, a, b, c))
        return fail(ctx, MI355_E_UNSUPPORTED, "lp256 GEMM: shape/layout not supported by this kernel");
    gemm_args g{};
    g.a = a; g.b = b; g.c = c;
    g.m = d.m; g.n = d.n; g.k = d.k;
    g.lda = d.lda; g.ldb = d.ldb; g.ldc = d.ldc;
    g.stride_a = d.stride_a; g.stride_b = d.stride_b; g.stride_c = d.stride_c;
    g.tiles_m = (uint32_t)((d.m + BM - 1) / BM);
    g.tiles_n = (uint32_t)((d.n + BN - 1) / BN);
    g.group_m = 8;
    const uint32_t batch = (uint32_t)d.batch;
    if (d.dtype_ab == MI355_DTYPE_BF16) {
        if (d.dtype_c == MI355_DTYPE_F32) launch<MI355_DTYPE_BF16, MI355_DTYPE_F32>(ctx, s, g, batch);
        else launch<MI355_DTYPE_BF16, MI355_DTYPE_BF16>(ctx, s, g, batch);
    } else {
        if (d.dtype_c == MI355_DTYPE_F32) launch<MI355_DTYPE_F16, MI355_DTYPE_F32>(ctx, s, g, batch);
        else launch<MI355_DTYPE_F16, MI355_DTYPE_F16>(ctx, s, g, batch);
    }
    check_launch(ctx, "mi355_gemm(lp256)");
    return MI355_OK;
}

}  // namespace mi355
