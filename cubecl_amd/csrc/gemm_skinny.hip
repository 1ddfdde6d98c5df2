// gemm_skinny.hip -- bf16 / f16 GEMM with at most 16 rows (or at most 16 columns): C[m][n] = sum_k A[m][k] * B[n][k].
//
// Roofline: HBM.  The large operand (8192 x 8192 bf16 = 128 MiB in the shape the bench quotes) is read exactly once and is
// all of the traffic; the small one (<= 16 rows, <= 256 KiB) stays in L2.  No matrix core: 16 rows would fill one MFMA
// operand and leave the kernel waiting on loads either way, and below 16 most of an MFMA tile would multiply zeros.  The
// arithmetic is v_dot2c_f32_{bf16,f16}: two 16-bit products and the running f32 sum per instruction.  Measured (bf16,
// N = K = 8192, 128 MiB streamed): M = 1 20.4 us = 6.57 TB/s, M = 2 21.8 us; the bf16 sum over the same 128 MiB takes 26 us
// and the split-K MFMA path this replaces 24.7 us.  From M = 3 up the per-lane dot products and the re-reads of the small
// operand (M x 16 KiB per wave, past the 32 KiB L1 at M = 4) cost more than the MFMA path's zero padding (M = 4: 27 us
// against 24.8, M = 16: 63 against 25.3), so AUTO takes this kernel for M <= 2 (or N <= 2) and wherever the MFMA kernels
// cannot run; forced, it serves up to 16 rows.
//
//   * one wave streams RW (1, 2 or 4) rows of the large operand, all 64 lanes along K: a lane's load is 16 bytes (8
//     elements), a wave's load 1 KiB of one row -- whole 128-byte lines, each read once, non-temporal.  Two K-chunks
//     (2 RW row loads per lane) are in flight per wave.
//   * the matching 16 bytes of each small-operand row are loaded once per K-chunk and used for all RW streamed rows.
//   * every lane ends with MT x RW partial sums over its K slices.  They are folded across the wave with a halving
//     butterfly (lane bit b: keep one half of the values, receive the other half's partner) -- V - 1 exchanges for V values
//     instead of 6 V -- after which lane i holds the finished value i; those lanes convert and store.
//
// Accumulation is f32 in a fixed order (K slices of a lane ascending, then the butterfly): deterministic, not bit-identical
// to the MFMA kernels (different association), within the parity tolerance of tests/test_gpu_gemm.py.
#include "gemm_common.hpp"

using namespace mi355;

namespace {

#ifndef SK_NT
#define SK_NT 1                  // dev: 0 = plain instead of non-temporal loads of the streamed operand (-3 ... -10 %)
#endif
// Streamed rows per wave (RW) and waves per workgroup, measured on 1 / 2 / 4 x 8192 x 8192 and 1 x 16384 x 16384
// (tools/dev/skinny_variants.py, us): RW 4 x 4 waves 26.1 / 26.4 / 30.0 / 94.0; RW 8 x 4: 36.3 / 34.6 / 45.0 / 102.5;
// RW 4 x 8: 32.0 / 29.5 / 36.2 / 85.7; RW 2 x 4: 20.9 / 22.2 / 27.0 / 89.1; RW 4 x 2: 21.5 / 21.8 / 28.5 / 82.2;
// RW 2 x 2: 21.2 / 22.3 / 27.3 / 79.1; RW 1 x 2: 20.4 / 23.7 / 37.4 / 79.9; RW 4 x 1: 21.3 / 21.5 / 27.5 / 80.0.
// Many small workgroups win: a wave's share is short (16-64 KiB), so what counts is how evenly the rows spread over the
// chip and how early every CU has loads in flight.  RW is chosen per row count of the small operand (launch_dt).
constexpr int WAVES = 2;         // waves per workgroup

__device__ __forceinline__ u32x4 stream_load(const void *p)
{
#if SK_NT
    return __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
#else
    return *reinterpret_cast<const u32x4 *>(p);
#endif
}

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

template <int DT>
__device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float acc)
{
    if constexpr (DT == MI355_DTYPE_BF16) {
        bf16x2 x, y;
        __builtin_memcpy(&x, &a, 4);
        __builtin_memcpy(&y, &b, 4);
        return __builtin_amdgcn_fdot2_f32_bf16(x, y, acc, false);
    } else {
        f16x2 x, y;
        __builtin_memcpy(&x, &a, 4);
        __builtin_memcpy(&y, &b, 4);
        return __builtin_amdgcn_fdot2(x, y, acc, false);
    }
}

// element traits: 16-bit operands go through dot2 (two products per instruction), f32 operands through plain FMAs
template <int DT> struct sk_elem { typedef uint16_t type; static constexpr int EPV = 8; };
template <> struct sk_elem<MI355_DTYPE_F32> { typedef float type; static constexpr int EPV = 4; };

// acc += the dot product of two 16-byte pieces
template <int DT>
__device__ __forceinline__ float dot16(const u32x4 &a, const u32x4 &b, float acc)
{
    if constexpr (DT == MI355_DTYPE_F32) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_fmaf(__uint_as_float(a[j]), __uint_as_float(b[j]), acc);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = dot2<DT>(a[j], b[j], acc);
    }
    return acc;
}

struct skinny_args {
    const void *small_;      // [small_rows][K], K contiguous
    const void *big;         // [big_rows][K], K contiguous
    void *out;
    int64_t ld_small, ld_big;            // elements
    int64_t out_stride_small, out_stride_big;   // elements between consecutive small / big indices of the output
    int64_t stride_small, stride_big, stride_out;   // batch strides, elements
    int32_t small_rows, big_rows;
    int32_t k;
    int32_t dtype_c;
};

__device__ __forceinline__ void store_one(void *out, int64_t idx, float v, int32_t dtype_c)
{
    if (dtype_c == MI355_DTYPE_F32) static_cast<float *>(out)[idx] = v;
    else if (dtype_c == MI355_DTYPE_BF16) static_cast<uint16_t *>(out)[idx] = f32_to_bf16_rne(v);
    else static_cast<uint16_t *>(out)[idx] = f32_to_f16_rne(v);
}

template <int DT, int MT, int RW>
__global__ __launch_bounds__(WAVES * 64) void gemm_skinny_kernel(skinny_args g)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row0 = ((int64_t)blockIdx.x * WAVES + wave) * RW;
    if (row0 >= g.big_rows) return;
    typedef typename sk_elem<DT>::type elem;
    constexpr int EPV = sk_elem<DT>::EPV, CH = 64 * EPV;                   // elements per 16-byte piece / per wave step
    const elem *small_ = static_cast<const elem *>(g.small_) + (int64_t)blockIdx.y * g.stride_small;
    const elem *big = static_cast<const elem *>(g.big) + (int64_t)blockIdx.y * g.stride_big;

    // rows past the end are clamped to the last one: loaded (harmlessly) and never stored
    const elem *brow[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) brow[r] = big + min(row0 + r, (int64_t)g.big_rows - 1) * g.ld_big;
    const elem *srow[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) srow[m] = small_ + (int64_t)min(m, g.small_rows - 1) * g.ld_small;

    float acc[MT][RW];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < RW; ++r) acc[m][r] = 0.0f;

    const int k_full = g.k / (2 * CH) * (2 * CH);
    int k0 = lane * EPV;
    // two chunks per trip: 2 x RW streamed loads in flight before the first dot product
    for (; k0 < k_full; k0 += 2 * CH) {
        u32x4 b0[RW], b1[RW];
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            b0[r] = stream_load(brow[r] + k0);
            b1[r] = stream_load(brow[r] + k0 + CH);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const u32x4 s0 = *reinterpret_cast<const u32x4 *>(srow[m] + k0);
            const u32x4 s1 = *reinterpret_cast<const u32x4 *>(srow[m] + k0 + CH);
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                acc[m][r] = dot16<DT>(s0, b0[r], acc[m][r]);
                acc[m][r] = dot16<DT>(s1, b1[r], acc[m][r]);
            }
        }
    }
    // remaining chunks, the last one possibly partial (K is a multiple of 8: a lane's 16 bytes are all in or all out)
    for (; k0 < g.k; k0 += CH) {
        u32x4 b0[RW];
#pragma unroll
        for (int r = 0; r < RW; ++r) b0[r] = stream_load(brow[r] + k0);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const u32x4 s0 = *reinterpret_cast<const u32x4 *>(srow[m] + k0);
#pragma unroll
            for (int r = 0; r < RW; ++r) acc[m][r] = dot16<DT>(s0, b0[r], acc[m][r]);
        }
    }

    // ---- fold the 64 lanes' partials: after the halving steps lane (i mod V) holds value i = m * RW + r -------------------
    constexpr int V = MT * RW;
    float v[V];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < RW; ++r) v[m * RW + r] = acc[m][r];
    int width = V;          // live values per lane
    int bit = 1;            // lane bit that decides which half a lane keeps
#pragma unroll
    for (int step = 0; step < 6; ++step, bit <<= 1) {
        if (width > 1) {
            const int half = width / 2;
            const bool upper = (lane & bit) != 0;
#pragma unroll
            for (int i = 0; i < V / 2; ++i) {
                if (i < half) {
                    const float keep = upper ? v[i + half] : v[i];
                    const float give = upper ? v[i] : v[i + half];
                    v[i] = keep + __shfl_xor(give, bit, 64);
                }
            }
            width = half;
        } else {
            v[0] += __shfl_xor(v[0], bit, 64);
        }
    }
    // which value did this lane end up with?  bit s of the value index (from the top) was chosen by lane bit s
    constexpr int LOGV = V == 1 ? 0 : V == 2 ? 1 : V == 4 ? 2 : V == 8 ? 3 : V == 16 ? 4 : V == 32 ? 5 : 6;
    int idx = 0;
#pragma unroll
    for (int s = 0; s < LOGV; ++s) idx |= ((lane >> s) & 1) << (LOGV - 1 - s);
    if (lane < V) {          // lanes >= V hold duplicates of the same values
        const int m = idx / RW, r = idx % RW;
        if (m < g.small_rows && row0 + r < g.big_rows) {
            char *out = static_cast<char *>(g.out);
            const int64_t o = (int64_t)blockIdx.y * g.stride_out + (int64_t)m * g.out_stride_small + (row0 + r) * g.out_stride_big;
            store_one(out, o, v[0], g.dtype_c);
        }
    }
}

template <int DT, int MT, int RW>
void launch_one(hipStream_t s, const skinny_args &g, uint32_t batch)
{
    const dim3 grid((uint32_t)((g.big_rows + WAVES * RW - 1) / (WAVES * RW)), batch), block(WAVES * 64);
    hipLaunchKernelGGL((gemm_skinny_kernel<DT, MT, RW>), grid, block, 0, s, g);
}

template <int DT>
void launch_dt(hipStream_t s, const skinny_args &g, uint32_t batch)
{
    if (g.small_rows <= 1) launch_one<DT, 1, 1>(s, g, batch);
    else if (g.small_rows <= 2) launch_one<DT, 2, 4>(s, g, batch);
    else if (g.small_rows <= 4) launch_one<DT, 4, 2>(s, g, batch);
    else if (g.small_rows <= 8) launch_one<DT, 8, 2>(s, g, batch);
    else launch_one<DT, 16, 2>(s, g, batch);
}

}  // namespace

namespace mi355 {

// A [M][K] and B [N][K] both K-contiguous, 16-bit or f32, one of M, N at most 16, 16-byte aligned rows, K a multiple of 8 (f32: 4).
bool gemm_skinny_supports(const mi355_gemm_desc &d, const void *a, const void *b, const void *c)
{
    (void)c;
    const bool f32 = d.dtype_ab == MI355_DTYPE_F32;          // f32 operands (round 4): f32 result only, plain FMAs instead of dot2
    if (d.dtype_ab != MI355_DTYPE_BF16 && d.dtype_ab != MI355_DTYPE_F16 && !f32) return false;
    if (f32 ? d.dtype_c != MI355_DTYPE_F32 : (d.dtype_c != MI355_DTYPE_F32 && d.dtype_c != MI355_DTYPE_BF16 && d.dtype_c != MI355_DTYPE_F16)) return false;
    if (d.trans_a || !d.trans_b) return false;
    const int64_t epv = f32 ? 4 : 8;                         // elements of a 16-byte piece
    if (d.m <= 0 || d.n <= 0 || d.k <= 0 || (d.k & (epv - 1)) || d.k > 0x7FFFFFF0) return false;
    if (d.m > 16 && d.n > 16) return false;
    if (d.m > 0x7FFFFFFF || d.n > 0x7FFFFFFF || d.batch < 1 || d.batch > 65535) return false;
    if ((d.lda & (epv - 1)) || (d.ldb & (epv - 1)) || (d.stride_a & (epv - 1)) || (d.stride_b & (epv - 1))) return false;
    if ((reinterpret_cast<uintptr_t>(a) & 15u) || (reinterpret_cast<uintptr_t>(b) & 15u)) return false;
    return true;
}

int32_t launch_gemm_skinny(mi355_ctx *ctx, hipStream_t s, const mi355_gemm_desc &d, const void *a, const void *b, void *c)
{
    if (!gemm_skinny_supports(d, a, b, c)) return fail(ctx, MI355_E_UNSUPPORTED, "mi355_gemm: the skinny kernel does not take this descriptor");
    skinny_args g{};
    const bool a_small = d.m <= d.n && d.m <= 16;   // otherwise N <= 16: the roles swap and the output is walked column-wise
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
    if (d.dtype_ab == MI355_DTYPE_F32) launch_dt<MI355_DTYPE_F32>(s, g, (uint32_t)d.batch);
    else if (d.dtype_ab == MI355_DTYPE_BF16) launch_dt<MI355_DTYPE_BF16>(s, g, (uint32_t)d.batch);
    else launch_dt<MI355_DTYPE_F16>(s, g, (uint32_t)d.batch);
    check_launch(ctx, "mi355_gemm(skinny)");
    return MI355_OK;
}

}  // namespace mi355
