// copy_strided.hip -- copy_into / into_contiguous / into_contiguous_packed: strided gathers into a (contiguous,
// pitched or itself strided) destination.
//
// Reference behaviour: crates/cubecl-std/src/tensor/contiguous/launch.rs:5-56 (into_contiguous, _pitched, copy_into),
// base.rs:295-389 (copy_gpu_ref: element q of the input's linear view -> element q of the output's linear layout),
// base.rs:170-293, :391-472 (the packed re-pack).  The reference generates ONE gather kernel whose vector width is
// the widest that both innermost axes allow and leaves everything else to the cache.  On MI355X the job is a pure
// HBM stream (2 x bytes), so the host first reduces the two views to the smallest joint iteration space
// (common refinement of the shapes, unit axes dropped, adjacent axes merged) and then picks the cheapest mover:
//
//   FLAT / ROWS   16-byte accesses on both sides; a row is located with a multiply-shift division per vector
//   TRANSPOSE     the input is contiguous along axis P, the output along another axis Q: a (256 B along P) x
//                 (256 B along Q) tile goes through LDS.  Sub-word elements are paired (2-byte) or quadrupled (1-byte)
//                 in registers with v_perm_b32 first, so LDS only ever sees dwords: tileT[p][q / R] with the dword
//                 column XOR-swizzled by (p / VE) -- the element-wise writes of a half-wave land in 32 different
//                 banks and the 16-byte row reads stay contiguous.  Both global sides move whole 256-byte segments
//                 with 16-byte accesses.
//   GENERIC       one element per access over the joint space (coalesced along the linear order)
//   TWO_SIDED     two strided views whose axis boundaries do not nest: each side decomposes the linear index by its own shape
//
// Roofline for all of them: HBM, 2 x elements x elem_size bytes per call.
#include <algorithm>
#include <vector>

#include "internal.hpp"

using namespace mi355;

namespace {

constexpr int MAXD = MI355_MAX_RANK * 2;   // a refinement can split axes: at most rank_in + rank_out - 1 pieces

// n / d for n < 2^32 by multiply-shift (Granlund & Montgomery): q = (mulhi(n, m) + n) >> s, the sum taken in 64 bits.
struct fdiv {
    uint32_t d, m, s, pad;
};

fdiv make_fdiv(uint64_t d64)
{
    fdiv f{};
    const uint32_t d = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(d64, 1), 0xFFFFFFFFull);
    uint32_t s = 0;
    while (s < 32 && (1ull << s) < d) ++s;
    f.d = d;
    f.s = s;
    f.m = (uint32_t)((((1ull << s) - d) << 32) / d + 1);
    return f;
}

__device__ __forceinline__ uint32_t fdiv_q(uint32_t n, const fdiv &f)
{
    return (uint32_t)(((uint64_t)__umulhi(n, f.m) + n) >> f.s);
}

// A joint (or one-sided) iteration space, innermost axis first.
struct dims {
    int32_t n;
    int32_t pad;
    uint64_t shape[MAXD];
    fdiv div[MAXD];
    int64_t s_in[MAXD];
    int64_t s_out[MAXD];
};

template <bool WIDE>
__device__ __forceinline__ void locate(uint64_t lin, const dims &d, int64_t &oi, int64_t &oo)
{
    oi = 0;
    oo = 0;
    if constexpr (WIDE) {
        uint64_t rem = lin;
        for (int i = 0; i < d.n - 1; ++i) {
            const uint64_t q = rem / d.shape[i], idx = rem - q * d.shape[i];
            oi += (int64_t)idx * d.s_in[i];
            oo += (int64_t)idx * d.s_out[i];
            rem = q;
        }
        oi += (int64_t)rem * d.s_in[d.n - 1];
        oo += (int64_t)rem * d.s_out[d.n - 1];
    } else {
        uint32_t rem = (uint32_t)lin;
        for (int i = 0; i < d.n - 1; ++i) {
            const uint32_t q = fdiv_q(rem, d.div[i]), idx = rem - q * d.div[i].d;
            oi += (int64_t)idx * d.s_in[i];
            oo += (int64_t)idx * d.s_out[i];
            rem = q;
        }
        oi += (int64_t)rem * d.s_in[d.n - 1];
        oo += (int64_t)rem * d.s_out[d.n - 1];
    }
}

// FLAT / ROWS / GENERIC: V is the access unit (16, 8, 4, 2 or 1 bytes); shapes and strides are in units of V.
// A workgroup walks whole tiles of 8 x 256 consecutive units: eight independent accesses per thread in flight, each
// of them coalesced across the wave.  STREAM (flat / rows: nothing is touched twice) marks both sides non-temporal.
template <typename V, bool WIDE, bool STREAM>
__global__ void __launch_bounds__(256)
joint_copy_kernel(const V *__restrict__ in, V *__restrict__ out, uint64_t total, dims d)
{
    constexpr int U = 8;
    const uint64_t tiles = (total + 256 * U - 1) / (256 * U);
    for (uint64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const uint64_t base = t * (256 * U) + threadIdx.x;
        V r[U];
        int64_t oo[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const uint64_t lin = base + k * 256;
            if (lin < total) {
                int64_t oi;
                if (d.n == 1) {
                    oi = (int64_t)lin * d.s_in[0];
                    oo[k] = (int64_t)lin * d.s_out[0];
                } else {
                    locate<WIDE>(lin, d, oi, oo[k]);
                }
                if constexpr (STREAM) r[k] = __builtin_nontemporal_load(in + oi);
                else r[k] = in[oi];
            }
        }
#pragma unroll
        for (int k = 0; k < U; ++k)
            if (base + k * 256 < total) {
                if constexpr (STREAM) __builtin_nontemporal_store(r[k], out + oo[k]);
                else out[oo[k]] = r[k];
            }
    }
}

// GENERIC with one side contiguous along the innermost joint axis: a thread moves K consecutive elements of that axis,
// one vector access on the contiguous side and K element accesses (stride s0) on the other.  One index decomposition
// per K elements instead of per element; `d` counts axis 0 in groups of K.
template <typename T, int K, bool GATHER, bool WIDE>
__global__ void __launch_bounds__(256)
pack_copy_kernel(const T *__restrict__ in, T *__restrict__ out, uint64_t groups, dims d, int64_t s0)
{
    typedef T vec __attribute__((ext_vector_type(K)));
    constexpr int U = 4;
    const uint64_t tiles = (groups + 256 * U - 1) / (256 * U);
    for (uint64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const uint64_t base = t * (256 * U) + threadIdx.x;
        vec r[U];
        int64_t oo[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const uint64_t lin = base + k * 256;
            if (lin < groups) {
                int64_t oi;
                if (d.n == 1) {
                    oi = (int64_t)lin * d.s_in[0];
                    oo[k] = (int64_t)lin * d.s_out[0];
                } else {
                    locate<WIDE>(lin, d, oi, oo[k]);
                }
                if constexpr (GATHER) {
#pragma unroll
                    for (int e = 0; e < K; ++e) r[k][e] = in[oi + e * s0];
                } else {
                    r[k] = *reinterpret_cast<const vec *>(in + oi);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < U; ++k)
            if (base + k * 256 < groups) {
                if constexpr (GATHER) {
                    *reinterpret_cast<vec *>(out + oo[k]) = r[k];
                } else {
#pragma unroll
                    for (int e = 0; e < K; ++e) out[oo[k] + e * s0] = r[k][e];
                }
            }
    }
}

// GENERIC, a SHORT axis P of NP = 2..4 elements that is innermost on one side ("interleaved": [.., q, p] packed) and a plane
// index on the other ("planar": q contiguous, p strided by `sp`): NHWC <-> NCHW images with few channels, complex <-> split
// re / im, RGB planes.  A thread moves K = 16 / sizeof(T) consecutive q for all NP planes: NP consecutive 16-byte vectors on
// the interleaved side, one 16-byte vector per plane on the planar side, and the NP x K elements change places in registers
// (byte permutes).  On the interleaved side a lane's NP vectors are consecutive, so the vectors additionally change LANES
// through 4 KiB of LDS per wave whenever the wave's 64 groups are one contiguous run there: every load / store instruction
// then moves 1 KiB whole instead of touching every NP-th vector.  `d` counts q in groups of K (axis 0) and carries the
// remaining axes; GATHER = the input is the interleaved side.  Every byte is moved once, in whole lines on both sides.
template <typename T, int NP, bool GATHER, bool WIDE>
__global__ void __launch_bounds__(256)
plane_copy_kernel(const T *__restrict__ in, T *__restrict__ out, uint64_t groups, dims d, int64_t sp)
{
    constexpr int K = 16 / (int)sizeof(T);
    typedef T vec __attribute__((ext_vector_type(K)));
    constexpr int U = 2;
    __shared__ vec stage[4 * NP * 64];                          // one wave's NP x 64 vectors at a time
    const uint64_t tiles = (groups + 256 * U - 1) / (256 * U);
    for (uint64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const uint64_t base = t * (256 * U) + threadIdx.x;
        vec a[U][NP];
        int64_t oo[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const uint64_t lin = base + k * 256;
            if (lin < groups) {
                int64_t oi;
                if (d.n == 1) {
                    oi = (int64_t)lin * d.s_in[0];
                    oo[k] = (int64_t)lin * d.s_out[0];
                } else {
                    locate<WIDE>(lin, d, oi, oo[k]);
                }
                if constexpr (GATHER) {
                    // the mirror image of the stores below: when the wave's 64 groups are one contiguous run of the input, every
                    // load instruction fetches 1 KiB whole and the vectors find their lanes through LDS
                    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
                    const int64_t first = ((int64_t)__builtin_amdgcn_readfirstlane((int)(oi >> 32)) << 32) |
                                          (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)oi);
                    if (__ballot(oi == first + (int64_t)lane * NP * K) == ~0ull) {
                        vec *st = stage + wave * (NP * 64);
#pragma unroll
                        for (int i = 0; i < NP; ++i)
                            st[i * 64 + lane] = __builtin_nontemporal_load(reinterpret_cast<const vec *>(in + first + ((int64_t)i * 64 + lane) * K));
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int i = 0; i < NP; ++i) a[k][i] = st[lane * NP + i];
                        __builtin_amdgcn_wave_barrier();
                        continue;
                    }
                }
#pragma unroll
                for (int i = 0; i < NP; ++i)
                    a[k][i] = __builtin_nontemporal_load(reinterpret_cast<const vec *>(in + oi + (GATHER ? (int64_t)i * K : (int64_t)i * sp)));
            }
        }
#pragma unroll
        for (int k = 0; k < U; ++k)
            if (base + k * 256 < groups) {
                vec r[NP];
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                    for (int e = 0; e < K; ++e) {
                        const int il = e * NP + pl;              // position in the interleaved run of NP x K elements
                        if constexpr (GATHER) r[pl][e] = a[k][il / K][il % K];
                        else r[il / K][il % K] = a[k][pl][e];
                    }
                if constexpr (!GATHER) {
                    // A lane's NP vectors are consecutive in the output, so store instruction i would put 16 bytes on every
                    // NP-th vector: each 128-byte line assembled from NP partial writes (measured: 2.2-3.9 TB/s against 4.9-5.4
                    // the other way).  When the wave's 64 groups are one contiguous run of the output -- always, except across
                    // the end of a row -- the vectors change lanes through LDS and every instruction stores 1 KiB whole.
                    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
                    const int64_t first = ((int64_t)__builtin_amdgcn_readfirstlane((int)(oo[k] >> 32)) << 32) |
                                          (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)oo[k]);
                    if (__ballot(oo[k] == first + (int64_t)lane * NP * K) == ~0ull) {
                        vec *st = stage + wave * (NP * 64);
#pragma unroll
                        for (int i = 0; i < NP; ++i) st[lane * NP + i] = r[i];
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int i = 0; i < NP; ++i) r[i] = st[i * 64 + lane];
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int i = 0; i < NP; ++i)
                            __builtin_nontemporal_store(r[i], reinterpret_cast<vec *>(out + first + ((int64_t)i * 64 + lane) * K));
                        continue;
                    }
                }
#pragma unroll
                for (int i = 0; i < NP; ++i)
                    __builtin_nontemporal_store(r[i], reinterpret_cast<vec *>(out + oo[k] + (GATHER ? (int64_t)i * sp : (int64_t)i * K)));
            }
    }
}

// TWO_SIDED: `di` decomposes the linear index by the input's (collapsed) shape, `dq` by the output's.
template <typename T, bool WIDE>
__global__ void __launch_bounds__(256)
two_sided_copy_kernel(const T *__restrict__ in, T *__restrict__ out, uint64_t total, dims di, dims dq)
{
    const uint64_t nthreads = (uint64_t)gridDim.x * 256;
    for (uint64_t lin = (uint64_t)blockIdx.x * 256 + threadIdx.x; lin < total; lin += nthreads) {
        int64_t oi, oo, unused;
        locate<WIDE>(lin, di, oi, unused);
        locate<WIDE>(lin, dq, unused, oo);
        out[oo] = in[oi];
    }
}

// ------------------------------------------------------------------------------------------------------------
// TRANSPOSE.  Joint axes: P (input stride 1), Q (output stride 1), and up to MAXD - 2 batch axes.
struct tr_args {
    uint64_t np, nq;          // extents of P and Q
    int64_t in_q, out_p;      // input stride along Q, output stride along P (elements)
    uint32_t tiles_p, tiles_q;
    uint32_t group, group_q, groups_p, groups_q;   // tile order: blocks of group (along P) x group_q (along Q) tiles
    int32_t nb;               // batch axes, innermost first
    int32_t vec_ok;           // all bases and strides are 16-byte multiples
    uint32_t bshape[MAXD];
    int64_t b_in[MAXD], b_out[MAXD];
};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#ifndef TR_SKEW
#define TR_SKEW 1   // dev: 0 = squares of tiles walked row by row
#endif
template <int ES> struct lds_elem;
template <> struct lds_elem<1> { typedef uint8_t type; };
template <> struct lds_elem<2> { typedef uint16_t type; };
template <> struct lds_elem<4> { typedef uint32_t type; };
template <> struct lds_elem<8> { typedef uint64_t type; };

template <int ES>
__global__ void __launch_bounds__(256)
transpose_copy_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, tr_args a)
{
    typedef typename lds_elem<ES>::type T;
    constexpr int VE = 16 / ES;        // elements per 16-byte access
    constexpr int R = ES >= 4 ? 1 : 4 / ES;   // q-rows packed into one LDS dword (an 8-byte element takes two dwords)
    constexpr int NJ = ES == 8 ? 32 : 64;     // row items of a tile: NJ x R rows of Q
    constexpr int TP = 16 * VE;        // tile extent along P: 256 bytes of input row
    constexpr int TQ = NJ * R;         // tile extent along Q: 256 bytes of output row
    __shared__ uint32_t lds[TP * 64];  // tileT[p][64 dword columns], column ^= ((p / VE) & 7) << 2

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // Tile order: G x GQ blocks of tiles, walked P-first inside a block and block by block along P (G = 1: plain
    // rows of tiles).  With 8 x 8 the ~2000 workgroups in flight read 2 KiB of each of their input rows and write
    // 2 KiB of each of their output rows; the host picks it when the row strides are multiples of 4 KiB (see there).
    uint32_t b = blockIdx.x;
    const uint32_t G = a.group, GQ = a.group_q, lp = b % G;
    b /= G;
    const uint32_t lq = b % GQ;
    b /= GQ;
    const uint32_t gp = b % a.groups_p;
    b /= a.groups_p;
    uint32_t gq = b % a.groups_q;
    b /= a.groups_q;
    // squares in flight at the same time differ in gp only: their output rows would all start at the same column offset
    // (addresses equal up to the row pitch -- the same few HBM channels when the pitch is a large power of two).  Skewing gq
    // by gp walks the squares diagonally: reads and writes both spread over the column offsets.  A bijection per gp.
    // 16384^2 bf16 4.59 -> 4.81 TB/s, 8192 x 4096 8-byte 4.6 -> 5.1 (tools/dev/copy_probe.py --transpose, interleaved).
    if (TR_SKEW && G > 1) gq = (gq + gp) % a.groups_q;   // G > 1 = pitches that are multiples of 4 KiB (host); plain rows of tiles lose 3-8 % to the skew
    const uint32_t tp = gp * G + lp, tq = gq * GQ + lq;
    if (tp >= a.tiles_p || tq >= a.tiles_q) return;
    int64_t off_in = 0, off_out = 0;
    for (int i = 0; i < a.nb; ++i) {
        const uint32_t idx = b % a.bshape[i];
        b /= a.bshape[i];
        off_in += (int64_t)idx * a.b_in[i];
        off_out += (int64_t)idx * a.b_out[i];
    }
    const uint64_t p0 = (uint64_t)tp * TP, q0 = (uint64_t)tq * TQ;
    const T *src = reinterpret_cast<const T *>(in) + off_in + (int64_t)q0 * a.in_q + (int64_t)p0;    // [q][p]
    T *dst = reinterpret_cast<T *>(out) + off_out + (int64_t)p0 * a.out_p + (int64_t)q0;              // [p][q]

    const bool whole = p0 + TP <= a.np && q0 + TQ <= a.nq;
    if (a.vec_ok) {
        // 16-byte accesses on both sides.  vec_ok also says both extents are multiples of VE, so at a ragged edge a
        // 16-byte piece is either entirely inside or entirely outside: `whole` tiles skip the tests.
        // load: item = (dword column j, 16-byte piece cv); lanes of a quad are 64 contiguous bytes, the next two lane
        // bits walk 4 neighbouring columns (bank bits 0-1), the rest walk cv (bank bits 2-4 through the swizzle)
        const int cv = (lane >> 4) * 4 + (lane & 3), jr = (lane >> 2) & 3;
        const bool p_in = whole || p0 + (uint64_t)(cv * VE) < a.np;
        u32x4 v[NJ / 16][R];
#pragma unroll
        for (int k = 0; k < NJ / 16; ++k) {
            const int j = (k * 4 + w) * 4 + jr;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (whole || (p_in && q0 + (uint64_t)(j * R + r) < a.nq))
                    v[k][r] = *reinterpret_cast<const u32x4 *>(src + (int64_t)(j * R + r) * a.in_q + cv * VE);
                else
                    v[k][r] = (u32x4){0u, 0u, 0u, 0u};
            }
        }
        const int swz = (cv & 7) << 2;
#pragma unroll
        for (int k = 0; k < NJ / 16; ++k) {
            const int j = (k * 4 + w) * 4 + jr;
            uint32_t *col = lds + (cv * VE) * 64 + ((ES == 8 ? 2 * j : j) ^ swz);
            if constexpr (ES == 8) {
#pragma unroll
                for (int e = 0; e < 2; ++e)
                    *reinterpret_cast<uint64_t *>(col + e * 64) = (uint64_t)v[k][0][2 * e] | ((uint64_t)v[k][0][2 * e + 1] << 32);
            } else if constexpr (ES == 4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) col[e * 64] = v[k][0][e];
            } else if constexpr (ES == 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    col[(2 * i) * 64] = __builtin_amdgcn_perm(v[k][1][i], v[k][0][i], 0x05040100u);
                    col[(2 * i + 1) * 64] = __builtin_amdgcn_perm(v[k][1][i], v[k][0][i], 0x07060302u);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t t0 = __builtin_amdgcn_perm(v[k][1][i], v[k][0][i], 0x05010400u);   // a0 b0 a1 b1
                    const uint32_t t1 = __builtin_amdgcn_perm(v[k][1][i], v[k][0][i], 0x07030602u);   // a2 b2 a3 b3
                    const uint32_t u0 = __builtin_amdgcn_perm(v[k][3][i], v[k][2][i], 0x05010400u);   // c0 d0 c1 d1
                    const uint32_t u1 = __builtin_amdgcn_perm(v[k][3][i], v[k][2][i], 0x07030602u);
                    col[(4 * i) * 64] = __builtin_amdgcn_perm(u0, t0, 0x05040100u);
                    col[(4 * i + 1) * 64] = __builtin_amdgcn_perm(u0, t0, 0x07060302u);
                    col[(4 * i + 2) * 64] = __builtin_amdgcn_perm(u1, t1, 0x05040100u);
                    col[(4 * i + 3) * 64] = __builtin_amdgcn_perm(u1, t1, 0x07060302u);
                }
            }
        }
        __syncthreads();
        // store: one wave writes 4 whole 256-byte output rows per pass
        const int c = lane & 15;
#pragma unroll
        for (int k = 0; k < TP / 16; ++k) {
            const int p = (k * 4 + w) * 4 + (lane >> 4);
            const u32x4 o = *reinterpret_cast<const u32x4 *>(lds + p * 64 + ((c * 4) ^ (((p / VE) & 7) << 2)));
            if (whole || (p0 + (uint64_t)p < a.np && q0 + (uint64_t)(c * VE) < a.nq))
                *reinterpret_cast<u32x4 *>(dst + (int64_t)p * a.out_p + c * VE) = o;
        }
    } else {
        // ragged or unaligned tile: the same LDS image, one element per access
        T *l = reinterpret_cast<T *>(lds);
        const auto at = [](int p, int q) {            // index of element (p, q) in the LDS image, in units of T
            const int swz = ((p / VE) & 7) << 2;
            if constexpr (ES == 8) return p * 32 + (((2 * q) ^ swz) >> 1);
            else return (p * 64 + ((q / R) ^ swz)) * R + (q % R);
        };
        const uint64_t pe = a.np - p0 < TP ? a.np - p0 : TP, qe = a.nq - q0 < TQ ? a.nq - q0 : TQ;
        for (int lin = tid; lin < TP * TQ; lin += 256) {
            const int q = lin / TP, p = lin % TP;
            if ((uint64_t)p < pe && (uint64_t)q < qe)
                l[at(p, q)] = src[(int64_t)q * a.in_q + p];
        }
        __syncthreads();
        for (int lin = tid; lin < TP * TQ; lin += 256) {
            const int p = lin / TQ, q = lin % TQ;
            if ((uint64_t)p < pe && (uint64_t)q < qe)
                dst[(int64_t)p * a.out_p + q] = l[at(p, q)];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// into_contiguous_packed.  One thread per output word; `lg` decomposes a logical element index by the logical
// shape (s_in = the input STORAGE strides in words), `dq` locates the output word.
template <typename W, bool WIDE>
__global__ void __launch_bounds__(256)
packed_copy_kernel(const W *__restrict__ in, W *__restrict__ out, uint64_t words, dims lg, dims dq, int32_t packed_axis,
                   uint32_t packing, uint32_t bits)
{
    const uint64_t nthreads = (uint64_t)gridDim.x * 256;
    const uint32_t mask = bits >= 32 ? 0xFFFFFFFFu : ((1u << bits) - 1u);
    for (uint64_t pos = (uint64_t)blockIdx.x * 256 + threadIdx.x; pos < words; pos += nthreads) {
        uint32_t acc = 0;
        for (uint32_t n = 0; n < packing; ++n) {
            uint64_t rem = pos * packing + n;
            int64_t off = 0;
            uint32_t slot = 0;
            for (int i = 0; i < lg.n; ++i) {                 // innermost first; the outermost axis wraps like the
                uint64_t q, idx;                             // reference's div_mod chain does (base.rs:186-195)
                if constexpr (WIDE) {
                    q = rem / lg.shape[i];
                    idx = rem - q * lg.shape[i];
                } else {
                    q = fdiv_q((uint32_t)rem, lg.div[i]);
                    idx = (uint32_t)rem - (uint32_t)q * lg.div[i].d;
                }
                rem = q;
                if (i == packed_axis) {
                    slot = (uint32_t)(idx % packing);
                    idx /= packing;
                }
                off += (int64_t)idx * lg.s_in[i];
            }
            const uint32_t word = (uint32_t)in[off];
            acc |= ((word >> (slot * bits)) & mask) << (n * bits);
        }
        int64_t oo, unused;
        locate<WIDE>(pos, dq, unused, oo);
        out[oo] = (W)acc;
    }
}

// ------------------------------------------------------------------------------------------------------------
// Host side: canonical form of the two views.
struct axis {
    uint64_t shape;
    int64_t si, so;
};

struct plan {
    int path = MI355_COPY_PATH_GENERIC;
    int access = 1;                 // bytes per access
    uint64_t total = 0;             // elements
    std::vector<axis> joint;        // innermost first; empty for TWO_SIDED
    std::vector<axis> in_only, out_only;
    int p_axis = -1, q_axis = -1;   // TRANSPOSE
    int pack_k = 1;                 // GENERIC: elements per vector access on the contiguous side (1: element-wise)
    bool gather = true;             // GENERIC with pack_k > 1: the OUTPUT is the contiguous side
    int planes = 0;                 // GENERIC: 2..4 = the short-axis mover (plane_copy_kernel); `gather` = the input is interleaved
    int q_axis_of_planes = 0;       //          joint axis that runs along the planes (0 or 1); the other of the two is the short one
};

const char *validate(const mi355_tensor_layout *l, const char *what, uint64_t &count)
{
    (void)what;
    if (!l) return "layout is NULL";
    if (l->rank < 1 || l->rank > MI355_MAX_RANK) return "rank must be 1..8";
    count = 1;
    for (int i = 0; i < l->rank; ++i) {
        if (l->shape[i] < 0) return "negative extent";
        if (l->strides[i] < 0) return "negative stride";
        if (l->shape[i] != 0 && count > (1ull << 62) / (uint64_t)l->shape[i]) return "element count overflows";
        count *= (uint64_t)l->shape[i];
    }
    return nullptr;
}

void merge(std::vector<axis> &v)
{
    std::vector<axis> r;
    for (const axis &x : v) {
        if (x.shape == 1) continue;
        if (!r.empty() && x.si == (int64_t)r.back().shape * r.back().si && x.so == (int64_t)r.back().shape * r.back().so)
            r.back().shape *= x.shape;
        else
            r.push_back(x);
    }
    if (r.empty()) r.push_back(axis{1, 1, 1});
    v.swap(r);
}

// One side on its own, innermost axis first, unit axes dropped and contiguous runs collapsed (so == si here).
std::vector<axis> collapse(const mi355_tensor_layout &l)
{
    std::vector<axis> v;
    for (int i = l.rank - 1; i >= 0; --i) v.push_back(axis{(uint64_t)l.shape[i], l.strides[i], l.strides[i]});
    merge(v);
    return v;
}

// Common refinement of the two (collapsed) shapes, walking both from the innermost axis: an axis of one side is cut
// wherever an axis boundary of the other side falls inside it.  Fails when a boundary does not divide ([2,3] against
// [3,2] of a strided source).
bool refine(const std::vector<axis> &a, const std::vector<axis> &b, std::vector<axis> &out)
{
    size_t i = 0, j = 0;
    uint64_t ra = a[0].shape, rb = b[0].shape;
    int64_t sa = a[0].si, sb = b[0].so;
    for (;;) {
        const uint64_t e = std::min(ra, rb);
        if (e == 0 || ra % e != 0 || rb % e != 0) return false;
        out.push_back(axis{e, sa, sb});
        ra /= e; sa *= (int64_t)e;
        rb /= e; sb *= (int64_t)e;
        if (ra == 1) { if (++i < a.size()) { ra = a[i].shape; sa = a[i].si; } }
        if (rb == 1) { if (++j < b.size()) { rb = b[j].shape; sb = b[j].so; } }
        if (i >= a.size() || j >= b.size()) return i >= a.size() && j >= b.size();
    }
}

int pow2_dividing(uint64_t x, int cap)
{
    int v = cap;
    while (v > 1 && x % (uint64_t)v != 0) v >>= 1;
    return v;
}

const char *make_plan(const void *in, const mi355_tensor_layout *li, const void *out, const mi355_tensor_layout *lo, int32_t es,
                      plan &pl)
{
    uint64_t ni = 0, no = 0;
    if (const char *e = validate(li, "input", ni)) return e;
    if (const char *e = validate(lo, "output", no)) return e;
    if (es != 1 && es != 2 && es != 4 && es != 8) return "elem_size must be 1, 2, 4 or 8";
    if (ni != no) return "the two views hold different numbers of elements";
    for (int i = 0; i < lo->rank; ++i)
        if (lo->strides[i] == 0 && lo->shape[i] > 1) return "the output view cannot broadcast";
    pl.total = ni;
    if (ni == 0) return nullptr;
    const uint64_t ain = (uint64_t)(uintptr_t)in, aout = (uint64_t)(uintptr_t)out;
    std::vector<axis> j;
    const std::vector<axis> ca = collapse(*li), cb = collapse(*lo);
    if (!refine(ca, cb, j)) {
        pl.path = MI355_COPY_PATH_TWO_SIDED;
        pl.access = es;
        pl.in_only = ca;
        pl.out_only = cb;
        return nullptr;
    }
    merge(j);
    if (j[0].si == 1 && j[0].so == 1) {
        int v = pow2_dividing(j[0].shape * (uint64_t)es, 16);
        v = pow2_dividing(ain, v);
        v = pow2_dividing(aout, v);
        for (size_t k = 1; k < j.size(); ++k) {
            v = pow2_dividing((uint64_t)j[k].si * (uint64_t)es, v);
            v = pow2_dividing((uint64_t)j[k].so * (uint64_t)es, v);
        }
        v = std::max(v, (int)es);
        // re-express in units of v bytes
        j[0].shape = j[0].shape * (uint64_t)es / (uint64_t)v;
        for (size_t k = 1; k < j.size(); ++k) {
            j[k].si = j[k].si * es / v;
            j[k].so = j[k].so * es / v;
        }
        pl.total = ni * (uint64_t)es / (uint64_t)v;
        pl.path = j.size() == 1 ? MI355_COPY_PATH_FLAT : MI355_COPY_PATH_ROWS;
        pl.access = v;
        pl.joint = j;
        return nullptr;
    }
    pl.joint = j;
    pl.access = es;
    pl.path = MI355_COPY_PATH_GENERIC;
    {
        int p = -1, q = -1;
        for (size_t k = 0; k < j.size(); ++k) {
            if (p < 0 && j[k].si == 1 && j[k].shape >= 16) p = (int)k;
            if (q < 0 && j[k].so == 1 && j[k].shape >= 16) q = (int)k;
        }
        if (p >= 0 && q >= 0 && p != q) {
            pl.path = MI355_COPY_PATH_TRANSPOSE;
            pl.p_axis = p;
            pl.q_axis = q;
            const uint64_t ve = 16 / (uint64_t)es;
            bool vec = ain % 16 == 0 && aout % 16 == 0 && j[p].shape % ve == 0 && j[q].shape % ve == 0;
            for (size_t k = 0; k < j.size(); ++k) {
                if ((int)k != p) vec = vec && ((uint64_t)j[k].si * (uint64_t)es) % 16 == 0;
                if ((int)k != q) vec = vec && ((uint64_t)j[k].so * (uint64_t)es) % 16 == 0;
            }
            pl.access = vec ? 16 : es;
            return nullptr;
        }
    }
    // a short axis (2..4 elements) innermost on one side, a plane index on the other: plane_copy_kernel
    if (es <= 4 && j.size() >= 2) {
        const int64_t K = 16 / es;
        // which of the two innermost joint axes is the short one depends on whose axis order the logical shape follows
        for (int combo = 0; combo < 4; ++combo) {
            const bool gather = (combo & 1) == 0;               // the input is the interleaved side
            const int qa = combo >> 1;                          // joint axis that runs along a plane
            const axis &Q = j[qa], &P = j[1 - qa];
            const int64_t q_planar = gather ? Q.so : Q.si, q_inter = gather ? Q.si : Q.so;
            const int64_t p_planar = gather ? P.so : P.si, p_inter = gather ? P.si : P.so;
            if (P.shape < 2 || P.shape > 4 || p_inter != 1 || q_planar != 1 || q_inter != (int64_t)P.shape) continue;
            if (Q.shape % (uint64_t)K != 0 || ain % 16 != 0 || aout % 16 != 0 || ((uint64_t)p_planar * (uint64_t)es) % 16 != 0) continue;
            bool ok = true;
            for (size_t k = 2; k < j.size(); ++k)
                ok = ok && ((uint64_t)j[k].si * (uint64_t)es) % 16 == 0 && ((uint64_t)j[k].so * (uint64_t)es) % 16 == 0;
            if (!ok) continue;
            pl.planes = (int)P.shape;
            pl.gather = gather;
            pl.q_axis_of_planes = qa;
            pl.access = 16;
            return nullptr;
        }
    }
    // one side contiguous along the innermost joint axis: K elements of it per thread
    if (es < 16 && (j[0].so == 1) != (j[0].si == 1)) {
        const bool gather = j[0].so == 1;
        const uint64_t addr = gather ? aout : ain;
        // How many elements per thread?  For one access of the element-wise side the lanes of a wave are K x s0 elements
        // apart when a row holds several groups, so a wide K scatters them over many cache lines (measured, 512 MiB,
        // stride-2 gather of 1-byte elements: K = 16 -> 1.2 TB/s, K = 4 -> 2.4 TB/s, K = 1 -> 1.7 TB/s).  When the
        // row is one group (or a few) and the next axis is the other side's contiguous one, the lanes walk that axis and
        // the accesses coalesce at any K ([*, 8, 8] transposes of 2-byte elements: K = 8 -> 5.2 TB/s, K = 2 -> 3.6 TB/s).
        const auto fit = [&](int cap) {
            int k = pow2_dividing(j[0].shape, cap);
            while (k > 1 && addr % (uint64_t)(k * es) != 0) k >>= 1;
            for (size_t a = 1; a < j.size(); ++a) k = pow2_dividing((uint64_t)(gather ? j[a].so : j[a].si), k);
            return k;
        };
        int k = fit(16 / es);
        const bool short_rows = j.size() >= 2 && j[0].shape <= 4 * (uint64_t)k && (gather ? j[1].si : j[1].so) == 1;
        if (!short_rows) k = fit(es == 1 ? 4 : es == 2 ? 4 : es == 4 ? 2 : 1);
        if (k >= 2) {
            pl.pack_k = k;
            pl.gather = gather;
            pl.access = k * es;
        }
    }
    return nullptr;
}

void fill_dims(dims &d, const std::vector<axis> &v)
{
    memset(&d, 0, sizeof(d));
    d.n = (int32_t)v.size();
    for (size_t k = 0; k < v.size(); ++k) {
        d.shape[k] = v[k].shape;
        d.div[k] = make_fdiv(v[k].shape);
        d.s_in[k] = v[k].si;
        d.s_out[k] = v[k].so;
    }
}

uint32_t stream_grid(const mi355_ctx *ctx, uint64_t items, int per_thread)
{
    const uint64_t blocks = (items + 256ull * per_thread - 1) / (256ull * per_thread);
    return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(blocks, (uint64_t)ctx->props.num_streaming_multiprocessors * 16));
}

template <typename V>
void launch_joint(hipStream_t s, uint32_t grid, bool wide, bool stream, const void *in, void *out, uint64_t total, const dims &d)
{
    if (wide)
        hipLaunchKernelGGL((joint_copy_kernel<V, true, false>), dim3(grid), dim3(256), 0, s, (const V *)in, (V *)out, total, d);
    else if (stream)
        hipLaunchKernelGGL((joint_copy_kernel<V, false, true>), dim3(grid), dim3(256), 0, s, (const V *)in, (V *)out, total, d);
    else
        hipLaunchKernelGGL((joint_copy_kernel<V, false, false>), dim3(grid), dim3(256), 0, s, (const V *)in, (V *)out, total, d);
}

template <typename T, int K>
void launch_pack_k(hipStream_t s, uint32_t grid, bool wide, bool gather, const void *in, void *out, uint64_t groups, const dims &d, int64_t s0)
{
    if (wide) {
        if (gather) hipLaunchKernelGGL((pack_copy_kernel<T, K, true, true>), dim3(grid), dim3(256), 0, s, (const T *)in, (T *)out, groups, d, s0);
        else hipLaunchKernelGGL((pack_copy_kernel<T, K, false, true>), dim3(grid), dim3(256), 0, s, (const T *)in, (T *)out, groups, d, s0);
    } else {
        if (gather) hipLaunchKernelGGL((pack_copy_kernel<T, K, true, false>), dim3(grid), dim3(256), 0, s, (const T *)in, (T *)out, groups, d, s0);
        else hipLaunchKernelGGL((pack_copy_kernel<T, K, false, false>), dim3(grid), dim3(256), 0, s, (const T *)in, (T *)out, groups, d, s0);
    }
}

template <typename T>
void launch_pack(hipStream_t s, uint32_t grid, bool wide, bool gather, int k, const void *in, void *out, uint64_t groups, const dims &d,
                 int64_t s0)
{
    constexpr int KMAX = 16 / (int)sizeof(T);
    if constexpr (KMAX >= 16) { if (k == 16) return launch_pack_k<T, 16>(s, grid, wide, gather, in, out, groups, d, s0); }
    if constexpr (KMAX >= 8) { if (k == 8) return launch_pack_k<T, 8>(s, grid, wide, gather, in, out, groups, d, s0); }
    if constexpr (KMAX >= 4) { if (k == 4) return launch_pack_k<T, 4>(s, grid, wide, gather, in, out, groups, d, s0); }
    launch_pack_k<T, 2>(s, grid, wide, gather, in, out, groups, d, s0);
}

template <typename T, int NP>
void launch_planes_np(hipStream_t s, uint32_t grid, bool wide, bool gather, const void *in, void *out, uint64_t groups, const dims &d, int64_t sp)
{
    if (wide) {
        if (gather) hipLaunchKernelGGL((plane_copy_kernel<T, NP, true, true>), dim3(grid), dim3(256), 0, s, (const T *)in, (T *)out, groups, d, sp);
        else hipLaunchKernelGGL((plane_copy_kernel<T, NP, false, true>), dim3(grid), dim3(256), 0, s, (const T *)in, (T *)out, groups, d, sp);
    } else {
        if (gather) hipLaunchKernelGGL((plane_copy_kernel<T, NP, true, false>), dim3(grid), dim3(256), 0, s, (const T *)in, (T *)out, groups, d, sp);
        else hipLaunchKernelGGL((plane_copy_kernel<T, NP, false, false>), dim3(grid), dim3(256), 0, s, (const T *)in, (T *)out, groups, d, sp);
    }
}

template <typename T>
void launch_planes(hipStream_t s, uint32_t grid, bool wide, bool gather, int np, const void *in, void *out, uint64_t groups, const dims &d, int64_t sp)
{
    if (np == 2) launch_planes_np<T, 2>(s, grid, wide, gather, in, out, groups, d, sp);
    else if (np == 3) launch_planes_np<T, 3>(s, grid, wide, gather, in, out, groups, d, sp);
    else launch_planes_np<T, 4>(s, grid, wide, gather, in, out, groups, d, sp);
}

template <typename T>
void launch_two_sided(hipStream_t s, uint32_t grid, bool wide, const void *in, void *out, uint64_t total, const dims &di,
                      const dims &dq)
{
    if (wide)
        hipLaunchKernelGGL((two_sided_copy_kernel<T, true>), dim3(grid), dim3(256), 0, s, (const T *)in, (T *)out, total, di, dq);
    else
        hipLaunchKernelGGL((two_sided_copy_kernel<T, false>), dim3(grid), dim3(256), 0, s, (const T *)in, (T *)out, total, di, dq);
}

}  // namespace

MI355_API int32_t mi355_copy_strided_plan(const void *in, const mi355_tensor_layout *in_layout, const void *out,
                                          const mi355_tensor_layout *out_layout, int32_t elem_size, int32_t *path,
                                          int32_t *access_bytes)
{
    plan pl;
    if (make_plan(in, in_layout, out, out_layout, elem_size, pl)) return MI355_E_INVALID_ARGUMENT;
    if (path) *path = pl.path;
    if (access_bytes) *access_bytes = pl.access;
    return MI355_OK;
}

MI355_API int32_t mi355_copy_strided(mi355_ctx *ctx, mi355_stream stream, const void *in, const mi355_tensor_layout *in_layout,
                                     void *out, const mi355_tensor_layout *out_layout, int32_t elem_size)
{
    MI355_REQUIRE_CTX(ctx);
    plan pl;
    if (const char *e = make_plan(in, in_layout, out, out_layout, elem_size, pl))
        return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_copy_strided: %s", e);
    if (pl.total == 0) return MI355_OK;
    if (!in || !out) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_copy_strided: NULL pointer");
    hipStream_t s = stream_of(ctx, stream);
    const bool wide = pl.total >= (1ull << 32);
    switch (pl.path) {
    case MI355_COPY_PATH_FLAT:
    case MI355_COPY_PATH_ROWS:
    case MI355_COPY_PATH_GENERIC: {
        dims d;
        if (pl.path == MI355_COPY_PATH_GENERIC && pl.planes > 0) {
            // axis 0 of the launch = q in groups of K (K elements on the planar side, K x planes on the interleaved one);
            // the short axis is walked inside the thread; every other joint axis as it is
            const int64_t K = 16 / elem_size;
            const axis &Q = pl.joint[pl.q_axis_of_planes], &P = pl.joint[1 - pl.q_axis_of_planes];
            std::vector<axis> g;
            g.push_back(axis{Q.shape / (uint64_t)K, Q.si * K, Q.so * K});
            for (size_t k = 2; k < pl.joint.size(); ++k) g.push_back(pl.joint[k]);
            fill_dims(d, g);
            const uint64_t groups = pl.total / (uint64_t)(K * pl.planes);
            const uint32_t grid = stream_grid(ctx, groups, 2);
            const int64_t sp = pl.gather ? P.so : P.si;
            const bool wide_g = groups >= (1ull << 32);
            switch (elem_size) {
            case 4: launch_planes<uint32_t>(s, grid, wide_g, pl.gather, pl.planes, in, out, groups, d, sp); break;
            case 2: launch_planes<uint16_t>(s, grid, wide_g, pl.gather, pl.planes, in, out, groups, d, sp); break;
            default: launch_planes<uint8_t>(s, grid, wide_g, pl.gather, pl.planes, in, out, groups, d, sp); break;
            }
            break;
        }
        if (pl.path == MI355_COPY_PATH_GENERIC && pl.pack_k > 1) {
            // axis 0 in groups of K elements: the contiguous side advances K per group, the other K x its stride
            std::vector<axis> g = pl.joint;
            const int64_t s0 = pl.gather ? g[0].si : g[0].so;
            g[0].shape /= (uint64_t)pl.pack_k;
            g[0].si *= pl.pack_k;
            g[0].so *= pl.pack_k;
            fill_dims(d, g);
            const uint64_t groups = pl.total / (uint64_t)pl.pack_k;
            const uint32_t grid = stream_grid(ctx, groups, 4);
            switch (elem_size) {
            case 8: launch_pack<uint64_t>(s, grid, wide, pl.gather, pl.pack_k, in, out, groups, d, s0); break;
            case 4: launch_pack<uint32_t>(s, grid, wide, pl.gather, pl.pack_k, in, out, groups, d, s0); break;
            case 2: launch_pack<uint16_t>(s, grid, wide, pl.gather, pl.pack_k, in, out, groups, d, s0); break;
            default: launch_pack<uint8_t>(s, grid, wide, pl.gather, pl.pack_k, in, out, groups, d, s0); break;
            }
            break;
        }
        fill_dims(d, pl.joint);
        const uint32_t grid = stream_grid(ctx, pl.total, 8);
        const bool stream = pl.path != MI355_COPY_PATH_GENERIC;
        switch (pl.access) {
        case 16: launch_joint<u32x4>(s, grid, wide, stream, in, out, pl.total, d); break;
        case 8: launch_joint<uint64_t>(s, grid, wide, stream, in, out, pl.total, d); break;
        case 4: launch_joint<uint32_t>(s, grid, wide, stream, in, out, pl.total, d); break;
        case 2: launch_joint<uint16_t>(s, grid, wide, stream, in, out, pl.total, d); break;
        default: launch_joint<uint8_t>(s, grid, wide, stream, in, out, pl.total, d); break;
        }
        break;
    }
    case MI355_COPY_PATH_TWO_SIDED: {
        dims di, dq;
        fill_dims(di, pl.in_only);
        fill_dims(dq, pl.out_only);
        const uint32_t grid = stream_grid(ctx, pl.total, 1);
        switch (elem_size) {
        case 8: launch_two_sided<uint64_t>(s, grid, wide, in, out, pl.total, di, dq); break;
        case 4: launch_two_sided<uint32_t>(s, grid, wide, in, out, pl.total, di, dq); break;
        case 2: launch_two_sided<uint16_t>(s, grid, wide, in, out, pl.total, di, dq); break;
        default: launch_two_sided<uint8_t>(s, grid, wide, in, out, pl.total, di, dq); break;
        }
        break;
    }
    case MI355_COPY_PATH_TRANSPOSE: {
        tr_args a;
        memset(&a, 0, sizeof(a));
        const axis &P = pl.joint[pl.p_axis], &Q = pl.joint[pl.q_axis];
        const int tp_ext = 256 / elem_size, tq_ext = elem_size == 8 ? 32 : 64 * (4 / elem_size);
        a.np = P.shape;
        a.nq = Q.shape;
        a.in_q = Q.si;
        a.out_p = P.so;
        const uint64_t tiles_p = (P.shape + tp_ext - 1) / tp_ext, tiles_q = (Q.shape + tq_ext - 1) / tq_ext;
        // Rows a multiple of 4 KiB apart on either side pile the pieces of one long row of tiles onto a few HBM channels
        // (measured, 16384^2 2-byte transpose: 4.2 TB/s walked row by row, 4.7 TB/s in 16 x 16 squares, 4.9-5.0 in 8 x 8); other
        // strides spread by themselves and prefer the plain row order (16640 x 15872: 5.1 against 4.9 TB/s).
#ifndef TR_GP
#define TR_GP 8    // blocks of 8 x 8 tiles: +2-4 % over 16 x 16 on 1-, 2- and 4-byte elements, level on 8-byte (32 x 32, 8 x 32, 64 x 4 ...: slower)
#endif
#ifndef TR_GQ
#define TR_GQ 8
#endif
        const bool camping = ((uint64_t)P.so * elem_size) % 4096 == 0 || ((uint64_t)Q.si * elem_size) % 4096 == 0;
        uint32_t group = camping ? TR_GP : 1, group_q = camping ? TR_GQ : 1;
        while (group > 1 && group > tiles_p) group >>= 1;
        while (group_q > 1 && group_q > tiles_q) group_q >>= 1;
        if (group == 1 || group_q == 1) group = group_q = 1;
        const uint64_t groups_p = (tiles_p + group - 1) / group, groups_q = (tiles_q + group_q - 1) / group_q;
        a.group = group;
        a.group_q = group_q;
        a.groups_p = (uint32_t)groups_p;
        a.groups_q = (uint32_t)groups_q;
        uint64_t blocks = groups_p * groups_q * group * group_q;
        a.vec_ok = pl.access == 16;
        for (size_t k = 0; k < pl.joint.size(); ++k) {
            if ((int)k == pl.p_axis || (int)k == pl.q_axis) continue;
            a.bshape[a.nb] = (uint32_t)pl.joint[k].shape;
            a.b_in[a.nb] = pl.joint[k].si;
            a.b_out[a.nb] = pl.joint[k].so;
            ++a.nb;
            blocks *= pl.joint[k].shape;
        }
        if (blocks >= (1ull << 31)) return fail(ctx, MI355_E_UNSUPPORTED, "mi355_copy_strided: %llu tiles", (unsigned long long)blocks);
        a.tiles_p = (uint32_t)tiles_p;
        a.tiles_q = (uint32_t)tiles_q;
        const uint8_t *ip = (const uint8_t *)in;
        uint8_t *op = (uint8_t *)out;
        if (elem_size == 4)
            hipLaunchKernelGGL(transpose_copy_kernel<4>, dim3((uint32_t)blocks), dim3(256), 0, s, ip, op, a);
        else if (elem_size == 8)
            hipLaunchKernelGGL(transpose_copy_kernel<8>, dim3((uint32_t)blocks), dim3(256), 0, s, ip, op, a);
        else if (elem_size == 2)
            hipLaunchKernelGGL(transpose_copy_kernel<2>, dim3((uint32_t)blocks), dim3(256), 0, s, ip, op, a);
        else
            hipLaunchKernelGGL(transpose_copy_kernel<1>, dim3((uint32_t)blocks), dim3(256), 0, s, ip, op, a);
        break;
    }
    default:
        return fail(ctx, MI355_E_EXECUTION, "mi355_copy_strided: no path");
    }
    check_launch(ctx, "mi355_copy_strided");
    return MI355_OK;
}

MI355_API int32_t mi355_copy_packed(mi355_ctx *ctx, mi355_stream stream, const void *in, const mi355_tensor_layout *in_storage,
                                    void *out, const mi355_tensor_layout *out_storage, const int64_t *shape, int32_t packed_dim,
                                    int32_t packing, int32_t word_size)
{
    MI355_REQUIRE_CTX(ctx);
    uint64_t ni = 0, no = 0;
    if (const char *e = validate(in_storage, "input", ni)) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_copy_packed: input %s", e);
    if (const char *e = validate(out_storage, "output", no)) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_copy_packed: output %s", e);
    if (!shape) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_copy_packed: shape is NULL");
    if (word_size != 4 && word_size != 1) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_copy_packed: word_size must be 4 or 1");
    const int word_bits = word_size * 8;
    if (packing < 1 || packing > word_bits || word_bits % packing != 0)
        return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_copy_packed: packing %d does not divide %d bits", packing, word_bits);
    const int rank = in_storage->rank;
    if (packed_dim < 0 || packed_dim >= rank) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_copy_packed: packed_dim %d of rank %d", packed_dim, rank);
    const int packed_axis_outer = rank - 1 - packed_dim;          // index into shape[], outermost first
    uint64_t want_words = 1;
    for (int i = 0; i < rank; ++i) {
        if (shape[i] < 0) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_copy_packed: negative extent");
        const uint64_t in_ext = i == packed_axis_outer ? ((uint64_t)shape[i] + packing - 1) / packing : (uint64_t)shape[i];
        if ((uint64_t)in_storage->shape[i] != in_ext)
            return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_copy_packed: input storage axis %d is %lld, the logical shape needs %llu", i,
                        (long long)in_storage->shape[i], (unsigned long long)in_ext);
        want_words *= i == rank - 1 ? ((uint64_t)shape[i] + packing - 1) / packing : (uint64_t)shape[i];
    }
    if (no != want_words)
        return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_copy_packed: the output holds %llu words, the re-packed tensor %llu",
                    (unsigned long long)no, (unsigned long long)want_words);
    if (no == 0) return MI355_OK;
    if (!in || !out) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_copy_packed: NULL pointer");
    // the logical decomposition keeps every axis (no merging: the packed axis divides differently)
    std::vector<axis> lg, oq;
    for (int i = rank - 1; i >= 0; --i) lg.push_back(axis{(uint64_t)std::max<int64_t>(shape[i], 1), in_storage->strides[i], 0});
    for (int i = out_storage->rank - 1; i >= 0; --i)
        oq.push_back(axis{(uint64_t)out_storage->shape[i], out_storage->strides[i], out_storage->strides[i]});
    merge(oq);
    dims dl, dq;
    fill_dims(dl, lg);
    fill_dims(dq, oq);
    const bool wide = no * (uint64_t)packing >= (1ull << 32);
    const uint32_t grid = stream_grid(ctx, no, 1);
    hipStream_t s = stream_of(ctx, stream);
    const uint32_t bits = (uint32_t)(word_bits / packing);
    const int32_t packed_axis = packed_dim;                        // innermost-first index
    if (word_size == 4) {
        if (wide)
            hipLaunchKernelGGL((packed_copy_kernel<uint32_t, true>), dim3(grid), dim3(256), 0, s, (const uint32_t *)in, (uint32_t *)out, no, dl, dq,
                               packed_axis, (uint32_t)packing, bits);
        else
            hipLaunchKernelGGL((packed_copy_kernel<uint32_t, false>), dim3(grid), dim3(256), 0, s, (const uint32_t *)in, (uint32_t *)out, no, dl, dq,
                               packed_axis, (uint32_t)packing, bits);
    } else {
        if (wide)
            hipLaunchKernelGGL((packed_copy_kernel<uint8_t, true>), dim3(grid), dim3(256), 0, s, (const uint8_t *)in, (uint8_t *)out, no, dl, dq,
                               packed_axis, (uint32_t)packing, bits);
        else
            hipLaunchKernelGGL((packed_copy_kernel<uint8_t, false>), dim3(grid), dim3(256), 0, s, (const uint8_t *)in, (uint8_t *)out, no, dl, dq,
                               packed_axis, (uint32_t)packing, bits);
    }
    check_launch(ctx, "mi355_copy_packed");
    return MI355_OK;
}
