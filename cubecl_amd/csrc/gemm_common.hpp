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
};

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
__device__ __forceinline__ void tile_coords(uint32_t lin, uint32_t tiles_m, uint32_t tiles_n, uint32_t group_m,
                                            uint32_t &tm, uint32_t &tn)
{
    const uint32_t per_group = group_m * tiles_n;
    const uint32_t group = lin / per_group;
    const uint32_t first_m = group * group_m;
    const uint32_t gsize = min(tiles_m - first_m, group_m);
    const uint32_t in_group = lin % per_group;
    tm = first_m + in_group % gsize;
    tn = in_group / gsize;
}

// Batched launches (grid.x = tiles, grid.y = batch; workgroups go to XCD (x + grid.x * y) % 8): the XCD remap runs over
// the whole batch-major tile sequence, so that the tiles resident on one XCD are a compact patch of ONE matrix of the
// batch (shared operand panels in that XCD's L2) rather than a few tiles each of several matrices.  Measured on the
// 256x256 kernel: +3 % at 64 x 2048^3, +8 % at 256 x 1024^3; identical to the per-matrix remap for batch == 1.
// Requires tiles * batch < 2^32 (checked by the kernels' supports()).
__device__ __forceinline__ void batched_tile_coords(uint32_t tiles_m, uint32_t tiles_n, uint32_t group_m, uint32_t &tm,
                                                    uint32_t &tn, uint32_t &batch)
{
    const uint32_t tiles_per = tiles_m * tiles_n;
    const uint32_t v = xcd_remap(blockIdx.y * tiles_per + blockIdx.x, tiles_per * gridDim.y);
    batch = v / tiles_per;
    tile_coords(v - batch * tiles_per, tiles_m, tiles_n, group_m, tm, tn);
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
int32_t launch_gemm_lp256w4(mi355_ctx *ctx, hipStream_t s, const mi355_gemm_desc &d, const void *a, const void *b, void *c,
                            const void *c_in = nullptr);
int32_t launch_gemm_lp256p(mi355_ctx *ctx, hipStream_t s, const mi355_gemm_desc &d, const void *a, const void *b, void *c);
bool gemm_lp256p_supports(const mi355_gemm_desc &d, const void *a, const void *b, const void *c);
int32_t launch_gemm_lp256q(mi355_ctx *ctx, hipStream_t s, const mi355_gemm_desc &d, const void *a, const void *b, void *c);
bool gemm_lp256q_supports(const mi355_gemm_desc &d, const void *a, const void *b, const void *c);
bool gemm_lp256w4_supports(const mi355_gemm_desc &d, const void *a, const void *b, const void *c);
int32_t launch_gemm_lp256x192(mi355_ctx *ctx, hipStream_t s, const mi355_gemm_desc &d, const void *a, const void *b, void *c, int tile_rows = 256);   // gemm_lp256w4.hip, NJ = 3 (tile_rows 192: NI = 3 too)
bool gemm_lp256x192_supports(const mi355_gemm_desc &d, const void *a, const void *b, const void *c);
int32_t launch_gemm_lp256m16(mi355_ctx *ctx, hipStream_t s, const mi355_gemm_desc &d, const void *a, const void *b, void *c);   // gemm_lp256m16.hip
bool gemm_lp256m16_supports(const mi355_gemm_desc &d, const void *a, const void *b, const void *c);
int32_t launch_gemm_lp256qm(mi355_ctx *ctx, hipStream_t s, const mi355_gemm_desc &d, const void *a, const void *b, void *c);    // gemm_lp256qm.hip
bool gemm_lp256qm_supports(const mi355_gemm_desc &d, const void *a, const void *b, const void *c);
int32_t launch_gemm_skinny(mi355_ctx *ctx, hipStream_t s, const mi355_gemm_desc &d, const void *a, const void *b, void *c);
bool gemm_skinny_supports(const mi355_gemm_desc &d, const void *a, const void *b, const void *c);
int32_t launch_gemm_nnrows(mi355_ctx *ctx, hipStream_t s, const mi355_gemm_desc &d, const void *a, const void *b, void *c);
bool gemm_nnrows_supports(const mi355_gemm_desc &d, const void *a, const void *b, const void *c);
bool gemm_nnrows_ready(mi355_ctx *ctx, hipStream_t s, const mi355_gemm_desc &d);   // its scratch exists or may be created now
int32_t launch_gemm_stream64(mi355_ctx *ctx, hipStream_t s, const mi355_gemm_desc &d, const void *a, const void *b, void *c);
bool gemm_stream64_supports(const mi355_gemm_desc &d, const void *a, const void *b, const void *c);
// ... its f32 form (gemm_stream64_f32.hip: v_mfma_f32_16x16x4_f32, both operands through per-wave LDS rings); reached through the two above
int32_t launch_gemm_stream64_f32(mi355_ctx *ctx, hipStream_t s, const mi355_gemm_desc &d, const void *a, const void *b, void *c);
bool gemm_stream64_f32_supports(const mi355_gemm_desc &d, const void *a, const void *b, const void *c);
int stream64_f32_blocks(const mi355_gemm_desc &d, int cus);
// block-scaled (MX) form of the same kernel; sa_t / sb_t are the re-arranged scales (gemm_scaled.hip)
bool gemm_lp256w4_mx_supports(const mi355_gemm_scaled_desc &d, const void *a, const void *b, const void *c);
int32_t launch_gemm_lp256w4_mx(mi355_ctx *ctx, hipStream_t s, const mi355_gemm_scaled_desc &d, const void *a, const void *sa_t,
                               int64_t stride_sa_t, const void *b, const void *sb_t, int64_t stride_sb_t, void *c);
// library-owned per-stream scratch + operand re-layout (gemm_relayout.hip)
enum { SCRATCH_SPLITK = 0, SCRATCH_RELAYOUT_A = 1, SCRATCH_RELAYOUT_B = 2, SCRATCH_SCALE_A = 3, SCRATCH_SCALE_B = 4, SCRATCH_PRODUCT = 5 };   // (6: reduce.hip's axis partials, 7: gemm_nnrows.hip's K-slice partials, 8: gemm_stream64.hip's)
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

}  // namespace mi355
