// gemm_common.hpp -- pieces shared by the GEMM kernels (device helpers + launch plumbing).
#pragma once

#include "internal.hpp"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

namespace mi355 {

// Division of a tile index by a launch constant.  The kernels turn a linear workgroup / tile number into (batch, tile row, tile
// column) with three or four unsigned divisions by values only the launch knows; as hipcc lowers them (v_rcp_iflag_f32 + two
// correction steps, ~40 dependent VALU instructions each) they were ~250 instructions in front of the first LDS-DMA of every
// workgroup, and 2 044 shader cycles of idle matrix pipe between two tiles of the persistent kernels (s_memtime stamps,
// profiles/r04_c5_counters.md).  The host therefore hands over a multiplier per divisor: q = umulhi(n, mul) >> shift, exact for
// n < 2^31 (mul = ceil(2^(31 + l) / d), l = ceil(log2 d), shift = l - 1; mul = 0 encodes d = 1).
struct fdiv_t { uint32_t mul, shift; };
inline fdiv_t make_fdiv(uint32_t d)
{
    if (d <= 1) return fdiv_t{0u, 0u};
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;                                  // l = ceil(log2 d) >= 1
    const uint64_t num = 1ull << (31 + l);
    return fdiv_t{(uint32_t)((num + d - 1) / d), l - 1};          // < 2^32 because 2^l < 2 d
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, uint32_t mul, uint32_t shift) { return mul ? (__umulhi(n, mul) >> shift) : n; }
#ifndef TILE_COORDS_DIVIDE
#define TILE_COORDS_DIVIDE 0   // dev (A/B builds only): 1 = the hardware-less division sequences of rounds 1-3 instead of the multipliers
#endif

struct gemm_args {
    const void *a;
    const void *b;
    void *c;
    int64_t m, n, k;
    int64_t lda, ldb, ldc;
    int64_t stride_a, stride_b, stride_c;
    uint32_t tiles_m, tiles_n;
    uint32_t group_m;
    uint32_t batch_count;    // persistent kernel: tiles x batch form one linear tile sequence
    uint32_t split_k;        // > 1: blockIdx.z selects a K slice and an f32 partial slab (lp128 split-K)
    int64_t split_c_stride;  // elements between the partial slabs of consecutive K slices
    const void *sa = nullptr, *sb = nullptr;   // MX: re-arranged ue8m0 scales ST[K-tile][padded row][blocks per K-tile row]
    int64_t stride_sa = 0, stride_sb = 0;      // bytes between batch entries of those
    const void *c_in = nullptr;                // f32 C only: D = A * B + c_in (same layout as c; may alias it), lp256w4
    uint32_t nt_mask = 0;                      // gemm_lp128.hip: bit 0 / 1 = the LDS-DMA pieces of A / B carry the non-temporal hint
    // multipliers for the tile-coordinate divisions (set_tile_divs; every supports() keeps tiles x batch below 2^31): by
    // tiles_m * tiles_n, by group_m * tiles_n and by tiles_m % group_m (the last, shorter group of tile rows); group_m itself is a
    // power of two.  Four dwords (the persistent kernels keep them in scalar registers across their tile loop):
    uint32_t fd_mul_tiles = 0, fd_mul_group = 0, fd_mul_tail = 0;
    uint32_t fd_shifts = 0;                    // bytes: shift of tiles | of group | of tail | log2(group_m)
};

// call once tiles_m, tiles_n, group_m (and the batch count) are final
inline void set_tile_divs(gemm_args &g, uint64_t batch)
{
    const uint64_t tiles = (uint64_t)g.tiles_m * g.tiles_n;
    (void)batch;
    uint32_t lg = 0;
    while ((1u << lg) < g.group_m) ++lg;
    if ((1u << lg) != g.group_m) { g.group_m = 1u << lg; }         // (every launcher passes 4 or 8)
    const fdiv_t t = make_fdiv((uint32_t)tiles), pg = make_fdiv(g.group_m * g.tiles_n),
                 tl = make_fdiv(g.tiles_m % g.group_m ? g.tiles_m % g.group_m : g.group_m);
    g.fd_mul_tiles = t.mul; g.fd_mul_group = pg.mul; g.fd_mul_tail = tl.mul;
    g.fd_shifts = t.shift | (pg.shift << 8) | (tl.shift << 16) | (lg << 24);
}

// MI355X dispatches workgroup b to XCD b % 8, each XCD with a private 4 MiB L2
// (MI355X_MICROARCH.md "Workgroup dispatch").  Remap the linear id so that every XCD walks one
// CONTIGUOUS chunk of the tile sequence (neighbouring tiles share A/B panels through that L2).
// Bijective for any nwg (guide section 5, "XCD swizzle must be bijective"); speed only.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t nwg)
{
    constexpr uint32_t NX = 8;
    const uint32_t q = nwg / NX, r = nwg % NX;
    const uint32_t xcd = bid % NX, local = bid / NX;
    const uint32_t base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + local;
}

// Grouped rasterisation: `group_m` tile-rows are swept column by column before moving down, so
// concurrently resident workgroups cover a compact block of the output.
//     group = lin / (group_m * tiles_n); gsize = min(tiles_m - group * group_m, group_m); in_group = lin % (group_m * tiles_n);
//     tm = group * group_m + in_group % gsize; tn = in_group / gsize           -- with the launch's multipliers (fdiv above)
__device__ __forceinline__ void tile_coords(uint32_t lin, const gemm_args &g, uint32_t &tm, uint32_t &tn)
{
#if TILE_COORDS_DIVIDE
    {
        const uint32_t per_group = g.group_m * g.tiles_n, group = lin / per_group, first_m = group * g.group_m;
        const uint32_t gsize = min(g.tiles_m - first_m, g.group_m), in_group = lin % per_group;
        tm = first_m + in_group % gsize; tn = in_group / gsize;
        return;
    }
#endif
    const uint32_t per_group = g.group_m * g.tiles_n;
    const uint32_t group = fdiv(lin, g.fd_mul_group, (g.fd_shifts >> 8) & 0xFFu);
    const uint32_t first_m = group * g.group_m;
    const uint32_t in_group = lin - group * per_group;
    const bool tail = g.tiles_m - first_m < g.group_m;             // the last, shorter group of tile rows
    const uint32_t gsize = tail ? g.tiles_m - first_m : g.group_m;
    tn = tail ? fdiv(in_group, g.fd_mul_tail, (g.fd_shifts >> 16) & 0xFFu) : in_group >> (g.fd_shifts >> 24);
    tm = first_m + in_group - tn * gsize;
}

// Batched launches (grid.x = tiles, grid.y = batch; workgroups go to XCD (x + grid.x * y) % 8): the XCD remap runs over
// the whole batch-major tile sequence, so that the tiles resident on one XCD are a compact patch of ONE matrix of the
// batch (shared operand panels in that XCD's L2) rather than a few tiles each of several matrices.  Measured on the
// 256x256 kernel: +3 % at 64 x 2048^3, +8 % at 256 x 1024^3; identical to the per-matrix remap for batch == 1.
// Requires tiles * batch < 2^32 (checked by the kernels' supports()).
__device__ __forceinline__ void batched_tile_coords(const gemm_args &g, uint32_t &tm, uint32_t &tn, uint32_t &batch)
{
    const uint32_t tiles_per = g.tiles_m * g.tiles_n;
    const uint32_t v = xcd_remap(blockIdx.y * tiles_per + blockIdx.x, tiles_per * gridDim.y);
    batch = TILE_COORDS_DIVIDE ? v / tiles_per : fdiv(v, g.fd_mul_tiles, g.fd_shifts & 0xFFu);
    tile_coords(v - batch * tiles_per, g, tm, tn);
}

__device__ __forceinline__ uint16_t f32_to_bf16_rne(float f)
{
    const __bf16 h = (__bf16)f;      // v_cvt_pk_bf16_f32 (gfx950), round-to-nearest-even
    uint16_t bits;
    __builtin_memcpy(&bits, &h, 2);
    return bits;
}

__device__ __forceinline__ uint16_t f32_to_f16_rne(float f)
{
    const _Float16 h = (_Float16)f;  // v_cvt_f16_f32, round-to-nearest-even
    uint16_t bits;
    __builtin_memcpy(&bits, &h, 2);
    return bits;
}

template <int DT>
__device__ __forceinline__ uint16_t f32_to_lp(float f)
{
    return DT == MI355_DTYPE_BF16 ? f32_to_bf16_rne(f) : f32_to_f16_rne(f);
}

// two values -> one dword of 16-bit results (lo in bits 0-15): one v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32 (via v_cvt_pkrtz is NOT used: RNE)
template <int DT>
__device__ __forceinline__ uint32_t f32x2_to_lp(float lo, float hi)
{
    uint32_t u;
    if constexpr (DT == MI355_DTYPE_BF16) {
        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
        const bf16x2_t v = {(__bf16)lo, (__bf16)hi};
        __builtin_memcpy(&u, &v, 4);
    } else {
        typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
        const f16x2_t v = {(_Float16)lo, (_Float16)hi};
        __builtin_memcpy(&u, &v, 4);
    }
    return u;
}

// host-side kernel launchers (one per translation unit)
int32_t launch_gemm_generic(mi355_ctx *ctx, hipStream_t s, const mi355_gemm_desc &d, const void *a, const void *b, void *c);
int32_t launch_gemm_f32_mfma(mi355_ctx *ctx, hipStream_t s, const mi355_gemm_desc &d, const void *a, const void *b, void *c);
int32_t launch_gemm_lp128(mi355_ctx *ctx, hipStream_t s, const mi355_gemm_desc &d, const void *a, const void *b, void *c);
int32_t launch_gemm_lp256x128(mi355_ctx *ctx, hipStream_t s, const mi355_gemm_desc &d, const void *a, const void *b, void *c);
int64_t lp128_split_count(const mi355_gemm_desc &d, int64_t cus);   // K slices of the 128x128 kernel's launcher (gemm_lp128.hip), 1 = none
bool gemm_lp256x128_supports(const mi355_gemm_desc &d, const void *a, const void *b, const void *c);
int32_t launch_gemm_lp256(mi355_ctx *ctx, hipStream_t s, const mi355_gemm_desc &d, const void *a, const void *b, void *c);
int32_t launch_gemm_lp256w4(mi355_ctx *ctx, hipStream_t s, const mi355_gemm_desc &d, const void *a, const void *b, void *c,
                            const void *c_in = nullptr);
int32_t launch_gemm_lp256p(mi355_ctx *ctx, hipStream_t s, const mi355_gemm_desc &d, const void *a, const void *b, void *c);
bool gemm_lp256p_supports(const mi355_gemm_desc &d, const void *a, const void *b, const void *c);
int32_t launch_gemm_lp256q(mi355_ctx *ctx, hipStream_t s, const mi355_gemm_desc &d, const void *a, const void *b, void *c);
bool gemm_lp256q_supports(const mi355_gemm_desc &d, const void *a, const void *b, const void *c);
bool gemm_lp256w4_supports(const mi355_gemm_desc &d, const void *a, const void *b, const void *c);
int32_t launch_gemm_skinny(mi355_ctx *ctx, hipStream_t s, const mi355_gemm_desc &d, const void *a, const void *b, void *c);
bool gemm_skinny_supports(const mi355_gemm_desc &d, const void *a, const void *b, const void *c);
int32_t launch_gemm_stream64(mi355_ctx *ctx, hipStream_t s, const mi355_gemm_desc &d, const void *a, const void *b, void *c);
bool gemm_stream64_supports(const mi355_gemm_desc &d, const void *a, const void *b, const void *c);
// block-scaled (MX) form of the same kernel; sa_t / sb_t are the re-arranged scales (gemm_scaled.hip)
bool gemm_lp256w4_mx_supports(const mi355_gemm_scaled_desc &d, const void *a, const void *b, const void *c);
int32_t launch_gemm_lp256w4_mx(mi355_ctx *ctx, hipStream_t s, const mi355_gemm_scaled_desc &d, const void *a, const void *sa_t,
                               int64_t stride_sa_t, const void *b, const void *sb_t, int64_t stride_sb_t, void *c);
// library-owned per-stream scratch + operand re-layout (gemm_relayout.hip)
enum { SCRATCH_SPLITK = 0, SCRATCH_RELAYOUT_A = 1, SCRATCH_RELAYOUT_B = 2, SCRATCH_SCALE_A = 3, SCRATCH_SCALE_B = 4, SCRATCH_PRODUCT = 5 };
int32_t scratch_get(mi355_ctx *ctx, hipStream_t s, int kind, size_t bytes, void **out);
void launch_transpose(hipStream_t s, const void *src, void *dst, int64_t rows, int64_t cols, int64_t ld_src, int64_t ld_dst,
                      int64_t batch, int64_t stride_src, int64_t stride_dst, int esz);
void launch_pad_copy(hipStream_t s, const void *src, void *dst, int64_t rows, int64_t cols, int64_t cols_pad, int64_t ld_src,
                     int64_t ld_dst, int64_t batch, int64_t stride_src, int64_t stride_dst, int esz);
// split-K plumbing (gemm_splitk.hip): per-stream library-owned f32 scratch + the slab fold kernel
int32_t splitk_scratch(mi355_ctx *ctx, hipStream_t s, size_t bytes, float **out);
void launch_splitk_fold(hipStream_t s, const float *slabs, uint32_t splits, int64_t slab_stride, int64_t batch, int64_t m,
                        int64_t n, void *c, int32_t dtype_c, int64_t ldc, int64_t stride_c);
// D = product + C, one rounding (gemm_add.hip)
void launch_add_c(hipStream_t s, const float *prod, const void *c_in, void *d_out, int64_t batch, int64_t m, int64_t n,
                  int32_t dtype_c, int64_t ldc, int64_t stride_c);
bool gemm_f32_mfma_supports(const mi355_gemm_desc &d, const void *a, const void *b, const void *c);
bool gemm_lp128_supports(const mi355_gemm_desc &d, const void *a, const void *b, const void *c);
bool gemm_lp256_supports(const mi355_gemm_desc &d, const void *a, const void *b, const void *c);

}  // namespace mi355
