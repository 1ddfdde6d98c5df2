// gemm_nnrows.hip -- at most 16 rows against a ROW-MAJOR weight: C[m][n] = sum_k A[m][k] * B[k][n], bf16 / f16 / f32, M <= 16.
//
// This is the layout `TensorHandle::new_contiguous` gives a rhs (crates/cubecl-std/src/tensor/handle.rs:89; a [K][N] weight
// classified RowMajor by matrix_batch_layout.rs:21-79): the decode-time product x [M][K] * W [K][N].  Roofline: HBM -- the
// weight (8192 x 8192 bf16 = 128 MiB in the shape the bench quotes) is read exactly once and is all of the traffic.
//
// What makes the layout awkward is that every matrix-core operand wants its K values contiguous per lane while a row of W
// is contiguous along N; the tile kernels transpose through LDS (gemm_lp128.hip BNN) and read W in 256-byte column strips,
// which HBM delivers at 5.1 TB/s at best (profiles/r03_hbm_colstrip_probe.txt).  This kernel never moves W through LDS:
//
//   * a workgroup owns a STRIP of S = 1024 / 512 / 256 bytes of every k-row of a K slice; a lane owns 8 columns (16 bytes of
//     a row) and loads FOUR consecutive k-rows of them -- one wave instruction reads S-byte pieces of 64 * 16 / S rows, whole
//     128-byte lines, non-temporal, U x 4 loads in flight per lane.  Wide strips with the slices of one strip on consecutive
//     workgroups stream at 5.6-5.8 TB/s (profiles/r04_hbm_rowslab_probe.txt).
//   * the 4 x 8 block a lane holds is transposed IN REGISTERS (16 v_perm_b32) into eight 4-long k-vectors, one per column,
//     which is exactly the B operand of v_mfma_f32_4x4x4_16b_{bf16,f16}: sixteen independent 4 x 4 x 4 blocks per
//     instruction, block = 4 adjacent lanes, each lane supplying ITS OWN column and receiving D[0..3][its column].  The A
//     operand (x[4r + lane % 4][the lane's four k]) comes from an LDS copy of x's K slice: broadcast reads, no conflicts.
//     8 x ceil(M / 4) MFMAs of 8 cycles per 64 bytes a lane loads: 4 us of matrix pipe for M = 16 on the quoted shape.
//   * the 4 * 64 * 16 / S copies of the strip's partial sums (waves x row groups) meet in LDS in a fixed order; with more than
//     one K slice the f32 partials go to library scratch and the LAST workgroup of the strip to arrive (one ticket word per
//     strip, gemm_nnrows' share of the stream's ticket slot) adds them in slice order and writes C -- one launch,
//     deterministic bits, no float atomics.
//
// Two more forms of the same loop (end of round 4): 9-16 rows on 256-byte strips use v_mfma_f32_16x16x16 -- its operand layout
// (lane % 16 = column, lane / 16 = one of four k groups) is what the lanes hold there, one instruction per column slot covers
// all 16 rows and the wave's four row groups are added in the matrix pipe; f32 operands use v_mfma_f32_4x4x1_16b_f32 -- one k
// per instruction, each lane's own column, so a row-major f32 weight needs no transposition at all (four columns per lane).
//
// Accumulation: f32, order fixed by (S, slices) = by the shape and the device's CU count -- run-to-run bit-identical; not
// bit-identical to the other kernels (different association), inside the parity tolerance of tests/test_gpu_gemm.py.
#include "gemm_common.hpp"
#include <type_traits>

using namespace mi355;

namespace {

#ifndef NNR_U
#define NNR_U 6                  // dev: four-row groups in flight per lane (4, 6 or 8)
#endif
// the rounds of the streaming loops written out (a `#pragma unroll` loop over the ring slots was not always unrolled, and a
// slot index that is not a constant turns the ring into register copies)
#if NNR_U == 4
#define NNR_EACH_U(X) X(0) X(1) X(2) X(3)
#elif NNR_U == 6
#define NNR_EACH_U(X) X(0) X(1) X(2) X(3) X(4) X(5)
#else
#define NNR_EACH_U(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#endif
typedef __attribute__((address_space(1))) const u32x4 *gptr_u32x4;
// Dev timing trace (-DNNR_TRACE, never in the product library): shader-clock stamps of thread 0 of every workgroup -- entry, x
// staged, K loop left, partials stored, ticket taken, exit -- read back with mi355_dev_nnr_trace (tools/dev/nnrows_trace.py).
#ifdef NNR_TRACE
__device__ unsigned long long nnr_trace_buf[1024 * 8];
#define NNR_STAMP(slot) do { if (threadIdx.x == 0 && blockIdx.x < 1024) nnr_trace_buf[blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define NNR_STAMP(slot)
#endif
constexpr int STRIP_TICKETS = 496;
constexpr int SCRATCH_NNROWS = 7;

struct nn_args {
    const void *a;               // x [M][K]
    const void *b;               // W [K][N]
    void *c;
    float *partial;              // [batch][slices][M][N] f32 when slices > 1
    unsigned int *tickets;       // [batch * strips], zero between calls
    int32_t m, n, k;
    int64_t lda, ldb, ldc, stride_a, stride_b, stride_c;
    int32_t strips, slices, ks;  // ks: k-rows per slice, a multiple of 64
    int32_t dtype_c;
};

// element traits: 16-bit operands (eight columns per 16-byte piece) or, since the end of round 4, f32 (four)
template <int DT> struct nn_elem { typedef uint16_t type; };
template <> struct nn_elem<MI355_DTYPE_F32> { typedef uint32_t type; };

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <int DT>
__device__ __forceinline__ f32x4 mfma4(u32x2 a, u32x2 b, f32x4 c)
{
    if constexpr (DT == MI355_DTYPE_BF16)
        return __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(h16x4, a), __builtin_bit_cast(h16x4, b), c, 0, 0, 0);
}

// v_mfma_f32_16x16x16_{bf16_1k,f16}: lane (j = lane % 16, g = lane / 16) supplies column j's (row i's for A) four k values
// 4 g ... 4 g + 3 and receives D[4 g ... 4 g + 3][j] -- with 256-byte strips (16 lanes per k-row, four row groups per wave) that
// is the layout the lanes already hold: ONE instruction per column slot covers 16 rows of x and adds the wave's four row
// groups inside the matrix pipe (the 4x4x4 form: four instructions and four separate partial sums).
template <int DT>
__device__ __forceinline__ f32x4 mfma16(u32x2 a, u32x2 b, f32x4 c)
{
    if constexpr (DT == MI355_DTYPE_BF16)
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(h16x4, a), __builtin_bit_cast(h16x4, b), c, 0, 0, 0);
}

__device__ __forceinline__ void store_c(void *c, int32_t dtype_c, int64_t off, float v)
{
    if (dtype_c == MI355_DTYPE_F32) static_cast<float *>(c)[off] = v;
    else if (dtype_c == MI355_DTYPE_BF16) static_cast<__bf16 *>(c)[off] = (__bf16)v;
    else static_cast<_Float16 *>(c)[off] = (_Float16)v;
}

template <int MB, int EB> struct nn_geom {
    static constexpr int MP = 4 * MB;                                      // x rows staged (zero rows past M)
    static constexpr int KC = 16384 / (MB * EB);                           // k values of x per LDS chunk (64 KiB): a chunk boundary drains the ring
};

template <int DT, int MB, int S>
__global__ void __launch_bounds__(256) gemm_nnrows_kernel(const nn_args g)
{
    constexpr int U = NNR_U;
    typedef typename nn_elem<DT>::type ET;
    constexpr int EB = sizeof(ET), EPV = 16 / EB;                          // bytes per element, columns per 16-byte piece
    constexpr bool F32 = DT == MI355_DTYPE_F32;
    constexpr int MP = nn_geom<MB, EB>::MP, KC = nn_geom<MB, EB>::KC;
    constexpr int LPR = S / 16, Q = 64 / LPR, COLS = S / EB;               // lanes per row piece, row groups per wave, columns per strip
    constexpr int RI = 16 * Q;                                             // k-rows one workgroup iteration covers
    constexpr int XP = KC + RI + 8;                                        // LDS pitch of an x row: the chunk + a zero tail + 16 B
    constexpr bool W16 = MB == 4 && S == 256 && !F32;                      // 9-16 rows on 256-byte strips: the 16x16x16 form (mfma16)
    constexpr int NCOPY = W16 ? 4 : 4 * Q;                                 // partial copies of the strip that meet in LDS
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ unsigned int is_last;
    ET *xs = reinterpret_cast<ET *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int p = lane % LPR, q = lane / LPR;
    const uint32_t strip = blockIdx.x / (uint32_t)g.slices, slice = blockIdx.x % (uint32_t)g.slices, batch = blockIdx.y;
    const int64_t col = (int64_t)strip * COLS + p * EPV;                   // my eight (f32: four) columns: all of them exist or none
    const ET *A = static_cast<const ET *>(g.a) + (int64_t)batch * g.stride_a;
    const int kbeg = (int)slice * g.ks, kend = min(g.k, kbeg + g.ks);

    NNR_STAMP(0);
    constexpr int AR = W16 ? 1 : MB;                                       // accumulator blocks per column slot
    f32x4 acc[AR][EPV];
#pragma unroll
    for (int r = 0; r < AR; ++r)
#pragma unroll
        for (int e = 0; e < EPV; ++e) acc[r][e] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // The W stream: a ring of U four-row groups per lane, refilled slot by slot.  What it takes for the compiler's waitcnt pass
    // to wait with vmcnt(4 (U - 1)) instead of draining the ring every round (each item measured in the ISA):
    //   * EVERY load is unconditional -- a load inside a branch makes the number in flight unknown, and the pass answers
    //     with vmcnt(0).  A slot with nothing left to fetch is refilled with 16 always-resident bytes (x itself), its address
    //     operands SELECTED, not branched over; a group that does not exist in a ragged last iteration is zeroed when consumed.
    //   * no load result may be dead on any path (x staging below: pieces past the end are stored to a dump slot) -- the
    //     pass guards every later write to such a register with a wait for the whole ring.
    //   * the slot index is a constant (rounds written out by macro: a `#pragma unroll` loop was not always unrolled, and an
    //     index that is not a constant turns the ring into register copies), and the refill is fenced behind the transposition
    //     that reads the slot (sched_barrier: the scheduler sinks all 4 U refills to the end of the round otherwise).
    // (Issuing the loads and the waits by hand in inline asm was tried: the register allocator may copy a slot's registers in
    // front of the hand-written wait -- it did, in the tail rounds -- and the copy reads a load that has not landed.)
    u32x4 v[U][4];
    const int64_t ldb2 = g.ldb * EB;                                       // bytes per k-row
    const uint32_t voff_lane = (uint32_t)(((int64_t)(w * Q + q) * 4 * g.ldb + (col < g.n ? p * EPV : 0)) * EB);
    const uint32_t voff_step = (uint32_t)((int64_t)RI * ldb2);
    const ET *dummy = A;                                              // always-resident 16 bytes (K >= EPV; x staging: loads that have nothing to fetch)
#ifdef NNR_ASM_LOADS      // dev: loads and waits by hand (see the note above: unsafe, kept for the record)
#define NNR_LOAD(dst, voff, sbase) asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory")
#define NNR_WAIT(n, u) asm volatile("s_waitcnt vmcnt(" #n ")" : "+v"(v[u][0]), "+v"(v[u][1]), "+v"(v[u][2]), "+v"(v[u][3])::"memory")
#else
#define NNR_LOAD(dst, voff, sbase) dst = __builtin_nontemporal_load(reinterpret_cast<gptr_u32x4>((sbase) + (uint64_t)(voff)))
#define NNR_WAIT(n, u)
#endif
    for (int kc0 = kbeg; kc0 < kend; kc0 += KC) {
        const int kc1 = min(kend, kc0 + KC);
        const int nit = (kc1 - kc0 + RI - 1) / RI;
        // group (it, w, q) = k-rows kc0 + ((it * 4 + w) * Q + q) * 4 ... + 3 (K % 8 == 0: a group exists whole or not at all)
        const char *sb0 = reinterpret_cast<const char *>(static_cast<const ET *>(g.b) + (int64_t)batch * g.stride_b + (int64_t)strip * COLS +
                                                         (int64_t)kc0 * g.ldb);
        const uint64_t sb[4] = {reinterpret_cast<uint64_t>(sb0), reinterpret_cast<uint64_t>(sb0 + ldb2), reinterpret_cast<uint64_t>(sb0 + 2 * ldb2),
                                reinterpret_cast<uint64_t>(sb0 + 3 * ldb2)};
        const int last_group = (kc1 - kc0) / 4 - 1;                        // lanes whose group of a ragged last iteration does not exist re-read this one
        // `real` false (wave-uniform): nothing left to fetch -- the slot is refilled all the same, with 16 always-resident bytes
        // (x itself), so that the number of loads in flight stays what the hand-written waits assume.  Address operands are
        // SELECTED, never branched over: two asm statements writing one slot in two arms would meet in a register copy.
        const uint64_t abase = reinterpret_cast<uint64_t>(A);
        auto issue = [&](bool real, int it, int u) __attribute__((always_inline)) {
            const int grp = min((it * 4 + w) * Q + q, last_group);
            const uint32_t voff = real ? voff_lane + (uint32_t)(grp - (w * Q + q)) * (uint32_t)(4 * ldb2) : 0u;
            NNR_LOAD(v[u][0], voff, real ? sb[0] : abase);
            NNR_LOAD(v[u][1], voff, real ? sb[1] : abase);
            NNR_LOAD(v[u][2], voff, real ? sb[2] : abase);
            NNR_LOAD(v[u][3], voff, real ? sb[3] : abase);
        };
        // a slot whose group has landed (the caller's wait) is transposed, refilled with the group U iterations ahead, multiplied
        auto consume = [&](auto steady, auto refill, int it, int u) __attribute__((always_inline)) {
            const int kk = ((it * 4 + w) * Q + q) * 4;
            if constexpr (!decltype(steady)::value) {
                if (kc0 + kk >= kc1) {                                     // ragged last iteration: my group does not exist
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[u][j] = (u32x4){0u, 0u, 0u, 0u};
                }
            }
            if constexpr (F32) {
                // f32: nothing to transpose -- v_mfma_f32_4x4x1_16b_f32 takes ONE k per instruction: each lane supplies its own column's
                // element of k-row j (B) and x[4 r + lane % 4][that row] (A), sixteen independent 4 x 4 blocks as in the 16-bit form
                const u32x4 w0 = v[u][0], w1 = v[u][1], w2 = v[u][2], w3 = v[u][3];
                if constexpr (decltype(refill)::value) {
                    __builtin_amdgcn_sched_barrier(0);
                    issue(decltype(steady)::value || it + U < nit, it + U, u);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (decltype(steady)::value || it < nit) {
#pragma unroll
                    for (int r = 0; r < MB; ++r) {
                        const u32x4 xa = *reinterpret_cast<const u32x4 *>(xs + (4 * r + (lane & 3)) * XP + kk);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            acc[r][e] = __builtin_amdgcn_mfma_f32_4x4x1f32(__uint_as_float(xa[0]), __uint_as_float(w0[e]), acc[r][e], 0, 0, 0);
                            acc[r][e] = __builtin_amdgcn_mfma_f32_4x4x1f32(__uint_as_float(xa[1]), __uint_as_float(w1[e]), acc[r][e], 0, 0, 0);
                            acc[r][e] = __builtin_amdgcn_mfma_f32_4x4x1f32(__uint_as_float(xa[2]), __uint_as_float(w2[e]), acc[r][e], 0, 0, 0);
                            acc[r][e] = __builtin_amdgcn_mfma_f32_4x4x1f32(__uint_as_float(xa[3]), __uint_as_float(w3[e]), acc[r][e], 0, 0, 0);
                        }
                    }
                }
            } else {
            // in-register transposition: column e of my eight -> its four k values
            u32x2 bv[8];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                bv[2 * d][0] = __builtin_amdgcn_perm(v[u][1][d], v[u][0][d], 0x05040100u);
                bv[2 * d][1] = __builtin_amdgcn_perm(v[u][3][d], v[u][2][d], 0x05040100u);
                bv[2 * d + 1][0] = __builtin_amdgcn_perm(v[u][1][d], v[u][0][d], 0x07060302u);
                bv[2 * d + 1][1] = __builtin_amdgcn_perm(v[u][3][d], v[u][2][d], 0x07060302u);
            }
            if constexpr (decltype(refill)::value) {
                __builtin_amdgcn_sched_barrier(0);
                issue(decltype(steady)::value || it + U < nit, it + U, u);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (decltype(steady)::value || it < nit) {                     // (wave-uniform; a round past the end only keeps the count)
                if constexpr (W16) {
                    const u32x2 xa = *reinterpret_cast<const u32x2 *>(xs + (lane & 15) * XP + kk);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[0][e] = mfma16<DT>(xa, bv[e], acc[0][e]);
                } else {
#pragma unroll
                    for (int r = 0; r < MB; ++r) {
                        const u32x2 xa = *reinterpret_cast<const u32x2 *>(xs + (4 * r + (lane & 3)) * XP + kk);
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[r][e] = mfma4<DT>(xa, bv[e], acc[r][e]);
                    }
                }
            }
            }
        };
        __syncthreads();                                                   // the previous chunk's x has been read
        {   // x[0 .. MP)[kc0 .. kc0 + nit * RI) -> LDS, zeros past M and past the slice; XB loads per thread in flight, unconditional.
            // The first batch of x goes out BEFORE the ring's first fill and is stored after it: queued behind 24 MiB of W
            // (all workgroups start together) x came back last, 10 000 cycles in, with the ring long landed and nothing in flight.
            constexpr int XB = 4 * MB;
            const int epr = nit * RI / EPV, pieces = MP * epr;             // 16-byte pieces per row / in all
            u32x4 xv[XB];
            auto x_load = [&](int base) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < XB; ++i) {
                    const int idx = base + i * 256 + tid, mm = idx / epr, kk = (idx - mm * epr) * EPV;
                    const bool ok = idx < pieces && mm < g.m && kc0 + kk < kc1;
                    uint64_t addr = ok ? reinterpret_cast<uint64_t>(A + (int64_t)mm * g.lda + kc0 + kk) : reinterpret_cast<uint64_t>(dummy);
                    asm("" : "+v"(addr));                                  // a select of ADDRESSES (else: a branch with a load in each arm)
                    xv[i] = *reinterpret_cast<gptr_u32x4>(addr);
                    if (!ok) xv[i] = (u32x4){0u, 0u, 0u, 0u};
                }
            };
            auto x_store = [&](int base) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < XB; ++i) {
                    const int idx = base + i * 256 + tid, mm = idx / epr, kk = (idx - mm * epr) * EPV;
                    // stored UNCONDITIONALLY (pieces past the end go to a dump slot): a load whose value one path never uses stays
                    // "in flight" for the compiler, which then drains vmcnt inside the streaming loop before reusing its registers
                    *reinterpret_cast<u32x4 *>(xs + (idx < pieces ? mm * XP + kk : MP * XP)) = xv[i];
                }
            };
            x_load(0);
            __builtin_amdgcn_sched_barrier(0);                            // (the scheduler would put the 4 U ring loads first again)
#define NNR_X(u) issue(u < nit, u, u);
            NNR_EACH_U(NNR_X)
#undef NNR_X
            __builtin_amdgcn_sched_barrier(0);
            x_store(0);
            for (int base = XB * 256; base < pieces; base += XB * 256) { x_load(base); x_store(base); }
        }
        __syncthreads();                                                   // (the compiler's vmcnt(0) for x also lands the first U groups)
        NNR_STAMP(1);
        int it0 = 0;
        // Every wait is a constant: before slot u is consumed, exactly U - 1 younger groups are in flight (4 loads each) in every
        // round but the last, which refills nothing (U - 1 - u younger).
#if NNR_U == 4
#define NNR_W_STEADY(u) NNR_WAIT(12, u)
#elif NNR_U == 6
#define NNR_W_STEADY(u) NNR_WAIT(20, u)
#else
#define NNR_W_STEADY(u) NNR_WAIT(28, u)
#endif
        // steady rounds: every consumed iteration exists whole and has a successor U ahead
#define NNR_X(u) NNR_W_STEADY(u); consume(std::true_type{}, std::true_type{}, it0 + u, u);
        for (; (it0 + 2 * U) * RI <= kc1 - kc0; it0 += U) { NNR_EACH_U(NNR_X) }
#undef NNR_X
        // rounds with a successor round: refills past the end fetch the resident bytes
#define NNR_X(u) NNR_W_STEADY(u); consume(std::false_type{}, std::true_type{}, it0 + u, u);
        for (; it0 + U < nit; it0 += U) { NNR_EACH_U(NNR_X) }
#undef NNR_X
        // the last round drains the ring
#define NNR_LAST(u, n) NNR_WAIT(n, u); consume(std::false_type{}, std::false_type{}, it0 + u, u);
#if NNR_U == 4
        NNR_LAST(0, 12) NNR_LAST(1, 8) NNR_LAST(2, 4) NNR_LAST(3, 0)
#elif NNR_U == 6
        NNR_LAST(0, 20) NNR_LAST(1, 16) NNR_LAST(2, 12) NNR_LAST(3, 8) NNR_LAST(4, 4) NNR_LAST(5, 0)
#else
        NNR_LAST(0, 28) NNR_LAST(1, 24) NNR_LAST(2, 20) NNR_LAST(3, 16) NNR_LAST(4, 12) NNR_LAST(5, 8) NNR_LAST(6, 4) NNR_LAST(7, 0)
#endif
#undef NNR_LAST
#undef NNR_W_STEADY
    }
#undef NNR_LOAD
#undef NNR_WAIT

    // ---- the 4 * Q copies of the strip meet in LDS (fixed order) --------------------------------------------------------------
    NNR_STAMP(2);
    __syncthreads();
    NNR_STAMP(3);
    float *red = reinterpret_cast<float *>(smem);                          // [NCOPY][MP][COLS]
    if constexpr (W16) {                                                   // lane (p, q) holds rows 4 q ... 4 q + 3 of its eight columns
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float *dst = red + ((w * MP + 4 * q + i) * COLS + p * 8);
            *reinterpret_cast<f32x4 *>(dst) = (f32x4){acc[0][0][i], acc[0][1][i], acc[0][2][i], acc[0][3][i]};
            *reinterpret_cast<f32x4 *>(dst + 4) = (f32x4){acc[0][4][i], acc[0][5][i], acc[0][6][i], acc[0][7][i]};
        }
    } else {
        const int copy = w * Q + q;
#pragma unroll
        for (int r = 0; r < MB; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float *dst = red + ((copy * MP + 4 * r + i) * COLS + p * EPV);
                *reinterpret_cast<f32x4 *>(dst) = (f32x4){acc[r][0][i], acc[r][1][i], acc[r][2][i], acc[r][3][i]};
                if constexpr (!F32) *reinterpret_cast<f32x4 *>(dst + 4) = (f32x4){acc[r][4][i], acc[r][5][i], acc[r][6][i], acc[r][7][i]};
            }
    }
    __syncthreads();
    const int64_t strip_col = (int64_t)strip * COLS;
    char *C = static_cast<char *>(g.c);
    const int64_t cbase = (int64_t)batch * g.stride_c;
    float *part = g.partial + ((int64_t)batch * g.slices) * g.m * g.n;    // [slices][M][N] of this batch entry
    typedef __attribute__((address_space(1))) unsigned long long gu64;
    typedef __attribute__((address_space(1))) unsigned int gu32;
    for (int idx = tid; idx < g.m * (COLS / 4); idx += 256) {
        const int mm = idx / (COLS / 4), c4 = (idx % (COLS / 4)) * 4;
        if (strip_col + c4 >= g.n) continue;
        f32x4 s = *reinterpret_cast<const f32x4 *>(red + (mm * COLS + c4));
#pragma unroll
        for (int cp = 1; cp < NCOPY; ++cp) s += *reinterpret_cast<const f32x4 *>(red + ((cp * MP + mm) * COLS + c4));
        if (g.slices == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) store_c(C, g.dtype_c, cbase + (int64_t)mm * g.ldc + strip_col + c4 + e, s[e]);
        } else {                                                           // write-through (sc1) 8-byte stores: see the hand-off below
            gu64 *dst = (gu64 *)reinterpret_cast<unsigned long long *>(part + ((int64_t)slice * g.m + mm) * g.n + strip_col + c4);
            __hip_atomic_store(dst, ((unsigned long long)__float_as_uint(s[1]) << 32) | __float_as_uint(s[0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(dst + 1, ((unsigned long long)__float_as_uint(s[3]) << 32) | __float_as_uint(s[2]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (g.slices == 1) return;

    // ---- K slices: the last workgroup of the strip to arrive adds the partials in slice order ---------------------------------
    // Hand-off as in reduce.hip: agent-scope 8-byte atomic stores (write-through) drained per wave, then the ticket; agent-scope
    // loads in the folding workgroup.  No fences: a release / acquire fence at agent scope is an L2 write-back / invalidate of the
    // whole XCD (buffer_wbl2 / buffer_inv sc1) -- with one per workgroup the kernel ran 50 us where the stream takes 24.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // my partials have left the wave ...
    __syncthreads();                                                       // ... every wave's have, before the ticket
    NNR_STAMP(4);
    unsigned int *ticket = g.tickets + batch * g.strips + strip;
    if (tid == 0) {
        const unsigned int old = __hip_atomic_fetch_add((gu32 *)ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = old == (unsigned int)g.slices - 1u ? 1u : 0u;
    }
    __syncthreads();
    NNR_STAMP(5);
    if (!is_last) return;
    // Sixteen 8-byte loads in flight per thread (one after the other, the same sum took 8 000 cycles for 16 rows x 4 slices:
    // every agent-scope load is a trip to memory): QD quads of the strip per thread and pass, J slices at a time -- one quad x
    // eight slices while the strip's quads fit one pass, two quads x four slices above.  Loads are unconditional (clamped
    // indices) so that they can all be issued; what does not exist is not added.  Slices are added in slice order.
    const int quads = g.m * (COLS / 4);
    const int64_t step = (int64_t)g.m * g.n / 2;                           // 8-byte words between slices (N % 8 == 0)
    auto fold = [&](auto qd_c, auto j_c) __attribute__((always_inline)) {
        constexpr int QD = decltype(qd_c)::value, J = decltype(j_c)::value;
        for (int base = 0; base < quads; base += QD * 256) {
            int mm[QD], c4[QD];
            bool live[QD];
            gu64 *src[QD];
#pragma unroll
            for (int qd = 0; qd < QD; ++qd) {
                const int idx = base + qd * 256 + tid;
                const int id = idx < quads ? idx : 0;
                mm[qd] = id / (COLS / 4); c4[qd] = (id % (COLS / 4)) * 4;
                live[qd] = idx < quads && strip_col + c4[qd] < g.n;
                if (!(strip_col + c4[qd] < g.n)) c4[qd] = 0;               // (a strip's first quad always exists)
                src[qd] = (gu64 *)reinterpret_cast<unsigned long long *>(part + (int64_t)mm[qd] * g.n + strip_col + c4[qd]);
            }
            f32x4 sum[QD];
#pragma unroll
            for (int qd = 0; qd < QD; ++qd) sum[qd] = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int sl0 = 0; sl0 < g.slices; sl0 += J) {
                unsigned long long wv[J][QD][2];
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    const int64_t off = (int64_t)min(sl0 + j, g.slices - 1) * step;
#pragma unroll
                    for (int qd = 0; qd < QD; ++qd) {
                        wv[j][qd][0] = __hip_atomic_load(src[qd] + off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        wv[j][qd][1] = __hip_atomic_load(src[qd] + off + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
#pragma unroll
                for (int j = 0; j < J; ++j)
                    if (sl0 + j < g.slices) {
#pragma unroll
                        for (int qd = 0; qd < QD; ++qd)
                            sum[qd] += (f32x4){__uint_as_float((uint32_t)wv[j][qd][0]), __uint_as_float((uint32_t)(wv[j][qd][0] >> 32)),
                                               __uint_as_float((uint32_t)wv[j][qd][1]), __uint_as_float((uint32_t)(wv[j][qd][1] >> 32))};
                    }
            }
#pragma unroll
            for (int qd = 0; qd < QD; ++qd)
                if (live[qd]) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) store_c(C, g.dtype_c, cbase + (int64_t)mm[qd] * g.ldc + strip_col + c4[qd] + e, sum[qd][e]);
                }
        }
    };
    if (quads <= 256) fold(std::integral_constant<int, 1>{}, std::integral_constant<int, 8>{});
    else fold(std::integral_constant<int, 2>{}, std::integral_constant<int, 4>{});
    if (tid == 0) __hip_atomic_store((gu32 *)ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next call
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    NNR_STAMP(6);
}

template <int DT, int MB, int S> constexpr size_t lds_bytes()
{
    constexpr size_t EB = sizeof(typename nn_elem<DT>::type);
    constexpr size_t x = (size_t)nn_geom<MB, EB>::MP * (nn_geom<MB, EB>::KC + 16 * (64 / (S / 16)) + 8) * EB + 16;   // + the dump slot
    constexpr size_t ncopy = (MB == 4 && S == 256 && EB == 2) ? 4 : 4 * (64 / (S / 16));
    constexpr size_t r = ncopy * nn_geom<MB, EB>::MP * (S / EB) * 4;      // the copies of the strip that meet in LDS
    return x > r ? x : r;
}

template <int DT, int MB, int S>
void launch_one(mi355_ctx *ctx, hipStream_t s, const nn_args &g, uint32_t batch)
{
    constexpr size_t LDS = lds_bytes<DT, MB, S>();
    lds_opt_in(ctx, reinterpret_cast<const void *>(gemm_nnrows_kernel<DT, MB, S>), LDS);
    hipLaunchKernelGGL((gemm_nnrows_kernel<DT, MB, S>), dim3((uint32_t)(g.strips * g.slices), batch), dim3(256), LDS, s, g);
}

template <int DT, int MB>
void launch_s(mi355_ctx *ctx, hipStream_t s, const nn_args &g, uint32_t batch, int strip_bytes)
{
    if (strip_bytes == 1024) launch_one<DT, MB, 1024>(ctx, s, g, batch);
    else if (strip_bytes == 512) launch_one<DT, MB, 512>(ctx, s, g, batch);
    else launch_one<DT, MB, 256>(ctx, s, g, batch);
}

template <int DT>
void launch_dt(mi355_ctx *ctx, hipStream_t s, const nn_args &g, uint32_t batch, int strip_bytes)
{
    static const int min_rows_16 = [] { const char *e = getenv("MI355_NNROWS_ROWS16_FROM"); return e ? atoi(e) : 9; }();   // dev: rows from which the 16-row form runs
    if (g.m >= min_rows_16 && strip_bytes == 256) launch_s<DT, 4>(ctx, s, g, batch, strip_bytes);
    else if (g.m <= 4) launch_s<DT, 1>(ctx, s, g, batch, strip_bytes);
    else if (g.m <= 8) launch_s<DT, 2>(ctx, s, g, batch, strip_bytes);
    else launch_s<DT, 4>(ctx, s, g, batch, strip_bytes);
}

// Strip width and K slices.  Measured (profiles/r04_nnrows_ab.txt, cold operands): the stream itself likes wide strips
// (1024 B: 23.8 us for 128 MiB against 25.7 at 256 B), but every K slice costs M x strip f32 partials written, read back and
// added by ONE workgroup per strip -- at 8192^2: M = 1 26.7 / 25.5 / 26.6 us with 1024 / 512 / 256-byte strips, M = 4
// 29.7 / 26.4 / 26.6, M = 16 45.0 / 32.9 / 31.0.  So: 512-byte strips up to four rows, 256-byte strips above; K cut so that
// at most one workgroup per CU exists (280 workgroups on 256 CUs ran 102 us where 224 run 80); a wider strip only where
// the narrow one would need more ticket words than a stream's slot holds.
struct nn_plan { int strip_bytes; int32_t strips, slices, ks; };

bool plan_for(const mi355_gemm_desc &d, int cus, nn_plan &out)
{
    static const int force_s = [] { const char *e = getenv("MI355_NNROWS_STRIP"); return e ? atoi(e) : 0; }();
    static const int force_slices = [] { const char *e = getenv("MI355_NNROWS_SLICES"); return e ? atoi(e) : 0; }();
    static const int per_cu = [] { const char *e = getenv("MI355_NNROWS_WG_PER_CU"); const int v = e ? atoi(e) : 1; return v > 0 ? v : 1; }();
    const int order_few[3] = {512, 256, 1024}, order_many[3] = {256, 512, 1024};
    for (int i = 0; i < 3; ++i) {
        const int sb = force_s ? force_s : (d.m <= 4 ? order_few[i] : order_many[i]);
        if (sb != 1024 && sb != 512 && sb != 256) return false;
        const int64_t strips = (d.n * (d.dtype_ab == MI355_DTYPE_F32 ? 4 : 2) + sb - 1) / sb;
        int64_t slices = std::max<int64_t>(1, (int64_t)cus * per_cu / (strips * d.batch));
        if (force_slices) slices = force_slices;
        slices = std::min<int64_t>(slices, (d.k + 63) / 64);
        const int64_t ks = ((d.k + slices - 1) / slices + 63) / 64 * 64;
        slices = (d.k + ks - 1) / ks;
        if (slices > 1 && strips * d.batch > STRIP_TICKETS) {
            if (force_s) return false;
            continue;
        }
        if (strips * slices * 1 > 0x7FFFFFFF) return false;
        out = {sb, (int32_t)strips, (int32_t)slices, (int32_t)ks};
        return true;
    }
    return false;
}

}  // namespace

MI355_API int32_t mi355_gemm_strip_plan(const mi355_gemm_desc *desc, int32_t compute_units, int32_t *out_strip_bytes, int32_t *out_strips,
                                        int32_t *out_slices)
{
    if (!desc || !out_strip_bytes || !out_strips || !out_slices || compute_units < 0) return MI355_E_INVALID_ARGUMENT;
    static const char aligned_dummy __attribute__((aligned(16))) = 0;
    *out_strip_bytes = *out_strips = *out_slices = 0;
    nn_plan p;
    if (!mi355::gemm_nnrows_supports(*desc, &aligned_dummy, &aligned_dummy, &aligned_dummy) || !plan_for(*desc, compute_units ? compute_units : 256, p))
        return MI355_OK;
    *out_strip_bytes = p.strip_bytes; *out_strips = p.strips; *out_slices = p.slices;
    return MI355_OK;
}

#ifdef NNR_TRACE
extern "C" __attribute__((visibility("default"))) int mi355_dev_nnr_trace(unsigned long long *host_out, int clear)
{
    const int rc = (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(nnr_trace_buf), sizeof(unsigned long long) * 1024 * 8);
    if (clear) { static unsigned long long z[1024 * 8]; (void)hipMemcpyToSymbol(HIP_SYMBOL(nnr_trace_buf), z, sizeof(z)); }
    return rc;
}
#endif

namespace mi355 {

// A [M][K] K-contiguous, B [K][N] row-major, both 16-bit with 16-byte aligned rows; M <= 16; N and K multiples of 8.
bool gemm_nnrows_supports(const mi355_gemm_desc &d, const void *a, const void *b, const void *c)
{
    (void)c;
    const bool f32 = d.dtype_ab == MI355_DTYPE_F32;                  // f32 operands: f32 result, four columns per 16-byte piece
    if (d.dtype_ab != MI355_DTYPE_BF16 && d.dtype_ab != MI355_DTYPE_F16 && !f32) return false;
    if (f32 ? d.dtype_c != MI355_DTYPE_F32 : (d.dtype_c != MI355_DTYPE_F32 && d.dtype_c != MI355_DTYPE_BF16 && d.dtype_c != MI355_DTYPE_F16)) return false;
    if (d.trans_a || d.trans_b) return false;
    const int64_t epv = f32 ? 4 : 8, eb = f32 ? 4 : 2;
    if (d.m < 1 || d.m > 16 || d.n < epv || d.k < epv || (d.n & (epv - 1)) || (d.k & (epv - 1))) return false;
    if (d.n > 0x3FFFFFF0 || d.k > 0x3FFFFFF0 || d.batch < 1 || d.batch > 65535) return false;
    if (d.ldb * eb * (16384 / eb + 72) >= (1ll << 32)) return false; // a lane's byte offset inside an x chunk (<= 16384 / eb + 64 k-rows) is 32-bit
    if ((d.lda & (epv - 1)) || (d.ldb & (epv - 1)) || (d.stride_a & (epv - 1)) || (d.stride_b & (epv - 1))) return false;
    if ((reinterpret_cast<uintptr_t>(a) & 15u) || (reinterpret_cast<uintptr_t>(b) & 15u)) return false;
    nn_plan p;
    return plan_for(d, 256, p);
}

// Outside a capture window everything can be allocated on the spot; inside one, the K-slice partials and the ticket words
// must already be there (a first call outside the window creates them).
bool gemm_nnrows_ready(mi355_ctx *ctx, hipStream_t s, const mi355_gemm_desc &d)
{
    if (!ctx->capturing) return true;
    nn_plan p;
    const int cus = ctx->props.num_streaming_multiprocessors > 0 ? ctx->props.num_streaming_multiprocessors : 256;
    if (!plan_for(d, cus, p)) return false;
    if (p.slices == 1) return true;
    if (!ctx->ticket_buf || ctx->tickets_dirty) return false;
    const auto it = ctx->scratch.find({s, SCRATCH_NNROWS});
    return it != ctx->scratch.end() && it->second.second >= (size_t)d.batch * p.slices * d.m * d.n * sizeof(float);
}

int32_t launch_gemm_nnrows(mi355_ctx *ctx, hipStream_t s, const mi355_gemm_desc &d, const void *a, const void *b, void *c)
{
    if (!gemm_nnrows_supports(d, a, b, c))
        return fail(ctx, MI355_E_UNSUPPORTED, "mi355_gemm: the few-rows x row-major-weight kernel does not take this descriptor");
    nn_plan p;
    const int cus = ctx->props.num_streaming_multiprocessors > 0 ? ctx->props.num_streaming_multiprocessors : 256;
    if (!plan_for(d, cus, p)) return fail(ctx, MI355_E_UNSUPPORTED, "mi355_gemm(nnrows): no strip plan");
    nn_args g{};
    g.a = a;
    g.b = b;
    g.c = c;
    g.m = (int32_t)d.m; g.n = (int32_t)d.n; g.k = (int32_t)d.k;
    g.lda = d.lda; g.ldb = d.ldb; g.ldc = d.ldc;
    g.stride_a = d.stride_a; g.stride_b = d.stride_b; g.stride_c = d.stride_c;
    g.strips = p.strips; g.slices = p.slices; g.ks = p.ks;
    g.dtype_c = d.dtype_c;
    if (p.slices > 1) {
        void *part = nullptr;
        const size_t bytes = (size_t)d.batch * p.slices * d.m * d.n * sizeof(float);
        if (scratch_get(ctx, s, SCRATCH_NNROWS, bytes, &part) != MI355_OK)
            return fail(ctx, MI355_E_UNSUPPORTED, "mi355_gemm(nnrows): no scratch for %zu bytes of partial sums (inside a capture window?)", bytes);
        g.partial = static_cast<float *>(part);
        const int32_t rc = strip_tickets_for_stream(ctx, s, &g.tickets);
        if (rc != MI355_OK) return rc;
    }
    if (d.dtype_ab == MI355_DTYPE_F32) launch_dt<MI355_DTYPE_F32>(ctx, s, g, (uint32_t)d.batch, p.strip_bytes);
    else if (d.dtype_ab == MI355_DTYPE_BF16) launch_dt<MI355_DTYPE_BF16>(ctx, s, g, (uint32_t)d.batch, p.strip_bytes);
    else launch_dt<MI355_DTYPE_F16>(ctx, s, g, (uint32_t)d.batch, p.strip_bytes);
    if (p.slices > 1 && hipPeekAtLastError() != hipSuccess) strip_tickets_mark_dirty(ctx);   // a refused launch never resets its tickets
    check_launch(ctx, "mi355_gemm(nnrows)");
    return MI355_OK;
}

}  // namespace mi355
