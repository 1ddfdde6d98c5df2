// gemm_lp256.hip -- bf16 / f16 GEMM, 256x256x64 workgroup tile, 8 waves in two "ping-pong" groups.
//
// Roofline: MFMA bf16/f16, ~2.5 PFLOP/s dense (MI355X_MICROARCH.md).  Why this shape: a 128x128
// tile needs 64 B/clk/CU of L2->LDS traffic at the MFMA rate (more than the ~56 B/clk/CU the L2
// delivers); 256x256x64 needs 32 B/clk/CU.
//
// Geometry
//   8 waves = 2 (M) x 4 (N); wave (g, wc) owns a 128 x 64 output = 4 x 2 MFMA tiles of 32x32
//   (v_mfma_f32_32x32x16, 128 accumulator registers).  g = wave >> 2 is also the ping-pong group:
//   waves w and w+4 share a SIMD, one from each group.
//   LDS: 2 stages x (A 256 rows + B 256 rows) x 128 B = 128 KiB, filled by LDS-DMA
//   (global_load_lds_dwordx4), rows of one 128-byte line each, XOR-swizzled on the SOURCE
//   address and on the fragment read exactly as in gemm_lp128.hip (conflict-free ds_read_b128).
//
// Schedule (per K-tile t, per wave): four phases, each {L: fragment reads + 2 DMA issues} barrier
//   {C: 8 MFMAs} barrier.  Group 1 runs one barrier behind group 0, so on every SIMD one wave is
//   in its MFMA segment while its partner does LDS reads and DMA issue (guide section 5, T3+T4+T5):
//       L1: read aLo (A rows 0-63 of my half), b0      issue Ahi(t+1)   C1: aLo x b0
//       L2: read b1                                    issue Alo(t+2)   C2: aLo x b1    + vmcnt(10)
//       L3: read aHi (rows 64-127)                     issue BX (t+2)   C3: aHi x b1
//       L4: -                                          issue BY (t+2)   C4: aHi x b0    + vmcnt(8)
//   A regions are private to a group (loaded and read by the same 4 waves); B is loaded by all 8
//   waves and freed after group 1's L2.  Every region is refilled at least one full phase after its
//   last read and at least 3 phases (~1700+ cycles) before its next read.  vmcnt never reaches 0 in
//   the loop: the waits are counted against the fixed issue order
//       ... Ahi(t) Alo(t+1) BX(t+1) BY(t+1) | Ahi(t+1) Alo(t+2) BX(t+2) BY(t+2) | ...
//   (2 DMA instructions per entry per wave).  Past the last K-tile the same instructions are
//   issued against a clamped tile index (harmless re-reads into dead regions) so the counts stay
//   uniform; the kernel drains them before its epilogue.
#include "gemm_common.hpp"

using namespace mi355;

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int ROW_BYTES = BK * 2;                 // 128
constexpr int OPER_BYTES = BM * ROW_BYTES;        // 32 KiB per operand per stage
constexpr int STAGE_BYTES = 2 * OPER_BYTES;       // 64 KiB
constexpr int LDS_BYTES = 2 * STAGE_BYTES;        // 128 KiB

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

#define WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
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

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;     // M half == ping-pong group
    const int wc = wave & 3;
    const int h = lane >> 5, l31 = lane & 31;

    uint32_t tm, tn;
    tile_coords(xcd_remap(blockIdx.x, g.tiles_m * g.tiles_n), g.tiles_m, g.tiles_n, g.group_m, tm, tn);
    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
    const int64_t batch = blockIdx.y;
    const char *__restrict__ A = static_cast<const char *>(g.a) + batch * g.stride_a * 2;
    const char *__restrict__ B = static_cast<const char *>(g.b) + batch * g.stride_b * 2;
    const int nk = (int)(g.k / BK);

    // ---- DMA source pointers (per lane) and LDS destinations (per wave) ------------------------
    // 8-row pieces of 1 KiB; lane -> (row = piece*8 + lane/8, physical chunk = lane%8)
    // Alo/Ahi: my group's A rows; this wave fills pieces wc*2 + j of the 8 pieces of each 64-row block
    // BX/BY  : B rows 0-127 / 128-255; this wave fills pieces wave*2 + j of each 16-piece block
    const char *src_alo[2], *src_ahi[2], *src_bx[2], *src_by[2];
    int dst_alo[2], dst_ahi[2], dst_bx[2], dst_by[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int sub = lane >> 3, c = lane & 7;
        {
            const int r = grp * 128 + (wc * 2 + j) * 8 + sub;            // A tile row (aLo block)
            const int q = c ^ ((r >> 1) & 7);
            src_alo[j] = A + (min(m0 + r, g.m - 1) * g.lda + q * 8) * 2;
            dst_alo[j] = (grp * 128 + (wc * 2 + j) * 8) * ROW_BYTES;
            const int r2 = r + 64;                                       // aHi block
            const int q2 = c ^ ((r2 >> 1) & 7);
            src_ahi[j] = A + (min(m0 + r2, g.m - 1) * g.lda + q2 * 8) * 2;
            dst_ahi[j] = dst_alo[j] + 64 * ROW_BYTES;
        }
        {
            const int r = (wave * 2 + j) * 8 + sub;                      // B tile row (BX block)
            const int q = c ^ ((r >> 1) & 7);
            src_bx[j] = B + (min(n0 + r, g.n - 1) * g.ldb + q * 8) * 2;
            dst_bx[j] = OPER_BYTES + (wave * 2 + j) * 8 * ROW_BYTES;
            const int r2 = r + 128;                                      // BY block
            const int q2 = c ^ ((r2 >> 1) & 7);
            src_by[j] = B + (min(n0 + r2, g.n - 1) * g.ldb + q2 * 8) * 2;
            dst_by[j] = dst_bx[j] + 128 * ROW_BYTES;
        }
    }
    auto issue = [&](const char *const (&src)[2], const int (&dst)[2], int tile, int stage) {
        const int64_t koff = (int64_t)min(tile, nk - 1) * (BK * 2);
        char *base = smem + stage * STAGE_BYTES;
#pragma unroll
        for (int j = 0; j < 2; ++j) glds16(src[j] + koff, base + dst[j]);
    };

    // ---- fragment read offsets (bytes within a stage) --------------------------------------------
    int ra[4], fa[4], rb[2], fb[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = grp * 128 + i * 32 + l31;
        ra[i] = row * ROW_BYTES; fa[i] = (row >> 1) & 7;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = wc * 64 + j * 32 + l31;
        rb[j] = OPER_BYTES + row * ROW_BYTES; fb[j] = (row >> 1) & 7;
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    frag ax[2][4];   // two A row-tiles x 4 k-steps (aLo, later aHi)
    frag b0[4], b1[4];

    auto read_a = [&](int stage, int first) {
        const char *base = smem + stage * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                ax[i][kk] = *reinterpret_cast<const frag *>(base + ra[first + i] + (((kk * 2 + h) ^ fa[first + i]) << 4));
    };
    auto read_b = [&](int stage, int j, frag (&dst)[4]) {
        const char *base = smem + stage * STAGE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
            dst[kk] = *reinterpret_cast<const frag *>(base + rb[j] + (((kk * 2 + h) ^ fb[j]) << 4));
    };
    auto compute = [&](int first, int j, const frag (&bf)[4]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[first + i][j] = lp<DT>::mfma(bf[kk], ax[i][kk], acc[first + i][j]);
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- prologue: tiles 0 and 1 in the steady-state issue order ----------------------------------
    issue(src_alo, dst_alo, 0, 0);
    issue(src_bx, dst_bx, 0, 0);
    issue(src_by, dst_by, 0, 0);
    issue(src_ahi, dst_ahi, 0, 0);
    issue(src_alo, dst_alo, 1, 1);
    issue(src_bx, dst_bx, 1, 1);
    issue(src_by, dst_by, 1, 1);
    WAIT_VMCNT(8);                       // everything through BY(0) has landed (this wave's share)
    PHASE_BARRIER();
    if (grp == 1) PHASE_BARRIER();       // group 1 runs one barrier behind

    for (int t = 0; t < nk; ++t) {
        const int cur = t & 1, nxt = cur ^ 1;
        // ---- phase 1 ----
        read_a(cur, 0);
        read_b(cur, 0, b0);
        issue(src_ahi, dst_ahi, t + 1, nxt);
        WAIT_LGKM0();
        PHASE_BARRIER();
        compute(0, 0, b0);
        PHASE_BARRIER();
        // ---- phase 2 ----
        read_b(cur, 1, b1);
        issue(src_alo, dst_alo, t + 2, cur);
        WAIT_VMCNT(10);                  // Ahi(t) landed: {Alo(t+1) BX(t+1) BY(t+1) Ahi(t+1) Alo(t+2)} may fly
        WAIT_LGKM0();
        PHASE_BARRIER();
        compute(0, 1, b1);
        PHASE_BARRIER();
        // ---- phase 3 ----
        read_a(cur, 2);
        issue(src_bx, dst_bx, t + 2, cur);
        WAIT_LGKM0();
        PHASE_BARRIER();
        compute(2, 1, b1);
        PHASE_BARRIER();
        // ---- phase 4 ----
        issue(src_by, dst_by, t + 2, cur);
        WAIT_VMCNT(8);                   // through BY(t+1): {Ahi(t+1) Alo(t+2) BX(t+2) BY(t+2)} may fly
        PHASE_BARRIER();
        compute(2, 0, b0);
        PHASE_BARRIER();
    }
    if (grp == 0) PHASE_BARRIER();       // re-align the barrier count of the two groups
    WAIT_VMCNT(0);                       // drain the clamped tail DMA before the workgroup retires

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
}

template <int DT, int DT_C>
void launch(mi355_ctx *ctx, hipStream_t s, const gemm_args &g, uint32_t batch, int slot)
{
    if (!(ctx->func_attr_mask & (1ull << slot))) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_lp256_kernel<DT, DT_C>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        ctx->func_attr_mask |= (1ull << slot);
    }
    hipLaunchKernelGGL((gemm_lp256_kernel<DT, DT_C>), dim3(g.tiles_m * g.tiles_n, batch), dim3(512), LDS_BYTES, s, g);
}

}  // namespace

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
    if (tiles > 0x7FFFFFFF) return false;
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
        if (d.dtype_c == MI355_DTYPE_F32) launch<MI355_DTYPE_BF16, MI355_DTYPE_F32>(ctx, s, g, batch, 8);
        else launch<MI355_DTYPE_BF16, MI355_DTYPE_BF16>(ctx, s, g, batch, 9);
    } else {
        if (d.dtype_c == MI355_DTYPE_F32) launch<MI355_DTYPE_F16, MI355_DTYPE_F32>(ctx, s, g, batch, 10);
        else launch<MI355_DTYPE_F16, MI355_DTYPE_F16>(ctx, s, g, batch, 11);
    }
    check_launch(ctx, "mi355_gemm(lp256)");
    return MI355_OK;
}

}  // namespace mi355
