/*
 * oracle.c -- CPU restatement of the tracel-ai/cubecl GEMM / reduce hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker / the reported CPU baseline.  The product path (libmi355cube.so) never links or
 * calls it and fails loudly when the HIP library is missing.
 *
 * Parity status
 * -------------
 * The reference cannot be compiled here (Rust; no cargo/rustc in the image) and its tiled
 * matmul / reduce kernel libraries live in the out-of-tree repo tracel-ai/cubek (reference
 * README.md:161-165), which is NOT a pinned dependency of the snapshot.  What the snapshot
 * does hold -- and what this file is pinned against in tests/test_oracle_golden.py -- are:
 *   - cmma fragment known-answer vectors
 *       crates/cubecl-core/src/runtime_tests/cmma.rs:552-576   (16x16x16 f16, Out = Lhs * Rhs^T)
 *       crates/cubecl-core/src/runtime_tests/cmma.rs:868-889   (16x16x8 "tf32", row-major B)
 *       crates/cubecl-core/src/runtime_tests/cmma.rs:932-1005  (strided lhs)
 *       crates/cubecl-core/src/runtime_tests/cmma.rs:695-722   (cube-scope expectation; the
 *                                                              loop oracle_gemm_* restates)
 *       crates/cubecl-core/src/runtime_tests/cmma.rs:1127-1177 (manual MMA, row-major A*B)
 *   - plane reductions: crates/cubecl-core/src/runtime_tests/plane.rs:154-190 and the
 *     xor-butterfly lowering crates/cubecl-cpp/src/shared/plane.rs:60-70
 *   - sequential sums: examples/sum_things/src/lib.rs:6-19 ([-1,10,1,5] -> 15),
 *     cubecl-book/src/getting-started/src/bin/v1-cpu.rs:7-15 (row sum)
 *   - collective closed form: crates/cubecl-core/src/runtime_tests/all_reduce.rs:52-59
 * Full-size tiled GEMM, array-wide sum and argmax have NO golden vector in the snapshot
 * ("parity unpinned" for those sizes): there the f64-accumulating functions
 * below are the oracle of record and the tolerance is BASELINE.json's (1e-5 relative for
 * f32 outputs, argmax indices bit-exact).
 *
 * Build: make -C oracle   (gcc -O2 -shared -fPIC -pthread)
 */
#include <math.h>
#include <pthread.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define ORACLE_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * Counter-based RNG shared bit-for-bit with the device fill kernel (cubecl_amd/csrc/fill.hip).
 * element i of tensor t uses counter (t, i), seed 0x5EEDC0BE.  splitmix64
 * finaliser over (seed, tensor, index); the top 24 bits become a uniform f32 in [0,1) --
 * every step is integer arithmetic or an exact int->float conversion, so host and device
 * agree exactly.
 * ------------------------------------------------------------------------------------------ */
static inline uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

static inline float rng_unit(uint64_t seed, uint64_t tensor, uint64_t i)
{
    uint64_t h = splitmix64(splitmix64(seed ^ (tensor * 0xD6E8FEB86659FD93ull)) + i);
    return (float)(uint32_t)(h >> 40) * (1.0f / 16777216.0f); /* 24 bits -> [0,1) exactly */
}

/* lo + (hi-lo)*u evaluated as one fmaf so the device (v_fma_f32) matches bitwise. */
ORACLE_API void oracle_fill_uniform_f32(float *dst, uint64_t n, uint64_t seed, uint64_t tensor,
                                        float lo, float hi)
{
    const float scale = hi - lo;
    for (uint64_t i = 0; i < n; ++i) dst[i] = fmaf(scale, rng_unit(seed, tensor, i), lo);
}

/* Elements [start, start + n) of the same stream: the RNG is counter-based (element i of tensor t = counter (t, i)),
 * so a window of a large device tensor can be regenerated without the elements in front of it. */
ORACLE_API void oracle_fill_uniform_f32_at(float *dst, uint64_t start, uint64_t n, uint64_t seed, uint64_t tensor,
                                           float lo, float hi)
{
    const float scale = hi - lo;
    for (uint64_t i = 0; i < n; ++i) dst[i] = fmaf(scale, rng_unit(seed, tensor, start + i), lo);
}

/* ------------------------------------------------------------------------------------------
 * 16-bit float conversions (no _Float16 / __bf16 in gcc 11 on x86).
 * ------------------------------------------------------------------------------------------ */
static inline uint32_t f32_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float bits_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* round-to-nearest-even, NaN kept quiet: half::bf16::from_f32 semantics */
ORACLE_API uint16_t oracle_f32_to_bf16(float f)
{
    uint32_t u = f32_bits(f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x0040u);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
ORACLE_API float oracle_bf16_to_f32(uint16_t h) { return bits_f32((uint32_t)h << 16); }

ORACLE_API uint16_t oracle_f32_to_f16(float f)
{
    uint32_t x = f32_bits(f);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t absx = x & 0x7FFFFFFFu;
    if (absx > 0x7F800000u) return (uint16_t)(sign | 0x7E00u);            /* NaN */
    if (absx >= 0x47800000u) return (uint16_t)(sign | 0x7C00u);           /* overflow -> inf */
    if (absx < 0x33000001u) return (uint16_t)sign;                         /* underflow -> 0 */
    int32_t exp = (int32_t)(absx >> 23) - 127;
    uint32_t man = (absx & 0x7FFFFFu) | 0x800000u;
    uint32_t shift, half;
    if (exp < -14) { shift = (uint32_t)(13 + (-14 - exp)); } else { shift = 13; }
    uint32_t q = man >> shift;
    uint32_t rem = man & ((1u << shift) - 1u);
    half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) q += 1;
    uint32_t out;
    if (exp < -14) out = q;                          /* subnormal (q may carry into exp=1) */
    else out = ((uint32_t)(exp + 15) << 10) + (q - 0x400u);
    return (uint16_t)(sign | out);
}
ORACLE_API float oracle_f16_to_f32(uint16_t h)
{
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu;
    uint32_t man = h & 0x3FFu;
    if (exp == 0) {
        if (man == 0) return bits_f32(sign);
        float v = (float)man * (1.0f / 16777216.0f); /* man * 2^-24 */
        return sign ? -v : v;
    }
    if (exp == 31) return bits_f32(sign | 0x7F800000u | (man << 13));
    return bits_f32(sign | ((exp + 112u) << 23) | (man << 13));
}

ORACLE_API void oracle_convert_f32_to_bf16(const float *src, uint16_t *dst, uint64_t n)
{ for (uint64_t i = 0; i < n; ++i) dst[i] = oracle_f32_to_bf16(src[i]); }
ORACLE_API void oracle_convert_f32_to_f16(const float *src, uint16_t *dst, uint64_t n)
{ for (uint64_t i = 0; i < n; ++i) dst[i] = oracle_f32_to_f16(src[i]); }
ORACLE_API void oracle_convert_bf16_to_f32(const uint16_t *src, float *dst, uint64_t n)
{ for (uint64_t i = 0; i < n; ++i) dst[i] = oracle_bf16_to_f32(src[i]); }
ORACLE_API void oracle_convert_f16_to_f32(const uint16_t *src, float *dst, uint64_t n)
{ for (uint64_t i = 0; i < n; ++i) dst[i] = oracle_f16_to_f32(src[i]); }

/* ------------------------------------------------------------------------------------------
 * 8-bit floats (OCP FP8).  The reference wraps the third-party `float8` crate (Cargo.lock: float8 0.7.0, not
 * vendored) in crates/cubecl-common/src/float/fp8/{fp8_e4m3,fp8_e5m2}.rs and states the contract there:
 *   e4m3 (fp8_e4m3.rs:12-37, :77-100): 1-4-3, bias 7, NO infinities, only S.1111.111 is NaN, MAX = 0x7E = 448;
 *   e5m2 (fp8_e5m2.rs:12-38, :78-100): 1-5-2, bias 15, IEEE-style: S.11111.00 = inf, S.11111.{01,10,11} = NaN,
 *                                      MAX = 0x7B = 57344;
 *   from_f32: round to nearest even; values too large, infinities included, SATURATE to +-MAX; NaN stays NaN;
 *             too-small values become subnormals or +-0.
 * Pinned by the known answers of those files' own tests (tests/test_oracle_golden.py).
 * ------------------------------------------------------------------------------------------ */
static inline uint8_t f32_to_fp8(float f, int mbits, int bias, uint8_t max_code)
{
    const uint32_t u = f32_bits(f);
    const uint8_t sign = (uint8_t)((u >> 24) & 0x80u);
    const uint32_t a = u & 0x7FFFFFFFu;
    if (a > 0x7F800000u) return (uint8_t)(sign | 0x7Fu);                    /* NaN */
    const float af = bits_f32(a);
    const float min_normal = ldexpf(1.0f, 1 - bias);
    uint32_t code;
    if (af < min_normal) {
        /* subnormal range: multiples of 2^(1-bias-mbits); rintf is round-to-nearest-even; the top value
         * 2^mbits is the encoding of the smallest normal, as it should be */
        code = (uint32_t)rintf(af * ldexpf(1.0f, bias - 1 + mbits));
    } else {
        const int shift = 23 - mbits;
        const uint32_t r = a + ((1u << (shift - 1)) - 1u) + ((a >> shift) & 1u);   /* RNE on the dropped bits */
        const uint32_t rebias = (uint32_t)(127 - bias) << mbits;
        code = (r >> shift) - rebias;
        if (r >= 0x7F800000u || code > max_code) code = max_code;           /* saturate (inf included) */
    }
    if (code > max_code) code = max_code;
    return (uint8_t)(sign | code);
}
ORACLE_API uint8_t oracle_f32_to_e4m3(float f) { return f32_to_fp8(f, 3, 7, 0x7E); }
ORACLE_API uint8_t oracle_f32_to_e5m2(float f) { return f32_to_fp8(f, 2, 15, 0x7B); }
ORACLE_API float oracle_e4m3_to_f32(uint8_t b)
{
    const uint32_t e = (b >> 3) & 15u, m = b & 7u;
    float v;
    if (e == 15u && m == 7u) return bits_f32(((uint32_t)(b & 0x80u) << 24) | 0x7FC00000u);
    if (e == 0u) v = (float)m * (1.0f / 512.0f);                            /* m * 2^-9 */
    else v = bits_f32(((e + 120u) << 23) | (m << 20));
    return (b & 0x80u) ? -v : v;
}
ORACLE_API float oracle_e5m2_to_f32(uint8_t b) { return oracle_f16_to_f32((uint16_t)((uint16_t)b << 8)); }
ORACLE_API void oracle_convert_f32_to_fp8(const float *src, uint8_t *dst, uint64_t n, int e5m2)
{ for (uint64_t i = 0; i < n; ++i) dst[i] = e5m2 ? oracle_f32_to_e5m2(src[i]) : oracle_f32_to_e4m3(src[i]); }
ORACLE_API void oracle_convert_fp8_to_f32(const uint8_t *src, float *dst, uint64_t n, int e5m2)
{ for (uint64_t i = 0; i < n; ++i) dst[i] = e5m2 ? oracle_e5m2_to_f32(src[i]) : oracle_e4m3_to_f32(src[i]); }

/* dtype codes shared with include/mi355cube.h (MI355_DTYPE_*) */
enum { DT_F32 = 0, DT_BF16 = 1, DT_F16 = 2, DT_F8E4M3 = 10, DT_F8E5M2 = 11 };

static inline float load_elem(const void *p, int dtype, int64_t idx)
{
    switch (dtype) {
    case DT_F32:  return ((const float *)p)[idx];
    case DT_BF16: return oracle_bf16_to_f32(((const uint16_t *)p)[idx]);
    case DT_F8E4M3: return oracle_e4m3_to_f32(((const uint8_t *)p)[idx]);
    case DT_F8E5M2: return oracle_e5m2_to_f32(((const uint8_t *)p)[idx]);
    default:      return oracle_f16_to_f32(((const uint16_t *)p)[idx]);
    }
}
static inline void store_elem(void *p, int dtype, int64_t idx, double v)
{
    switch (dtype) {
    case DT_F32:  ((float *)p)[idx] = (float)v; break;
    case DT_BF16: ((uint16_t *)p)[idx] = oracle_f32_to_bf16((float)v); break;
    case DT_F8E4M3: ((uint8_t *)p)[idx] = oracle_f32_to_e4m3((float)v); break;
    case DT_F8E5M2: ((uint8_t *)p)[idx] = oracle_f32_to_e5m2((float)v); break;
    default:      ((uint16_t *)p)[idx] = oracle_f32_to_f16((float)v); break;
    }
}

/* ------------------------------------------------------------------------------------------
 * GEMM.  Restates test_simple_cube_expected (runtime_tests/cmma.rs:695-722): inputs widened
 * to f32, `sum += lhs * rhs` sequentially over k in f32, row-major output.  trans_b != 0 is
 * the tests' `Out = Lhs * Rhs^T` form (B stored [N][K], "ColMajor" B, cmma.rs:23);
 * trans_b == 0 is the row-major A[m,k]*B[k,n] form of the manual-MMA test (cmma.rs:1160-1177).
 * Strides are in elements (TensorHandle convention, crates/cubecl-std/src/tensor/handle.rs).
 * batch strides may be 0 (broadcast operand, matrix_batch_layout.rs:21-79).
 * acc_f64 != 0 accumulates in double instead: the numerical oracle for the 1e-5 check.
 * ------------------------------------------------------------------------------------------ */
ORACLE_API void oracle_gemm(const void *A, const void *B, void *C, int dtype_ab, int dtype_c,
                            int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                            int64_t ldc, int trans_b, int64_t batch, int64_t stride_a,
                            int64_t stride_b, int64_t stride_c, int acc_f64)
{
    for (int64_t b = 0; b < batch; ++b) {
        const int64_t oa = b * stride_a, ob = b * stride_b, oc = b * stride_c;
        for (int64_t m = 0; m < M; ++m) {
            for (int64_t n = 0; n < N; ++n) {
                if (acc_f64) {
                    double sum = 0.0;
                    for (int64_t k = 0; k < K; ++k) {
                        double l = load_elem(A, dtype_ab, oa + m * lda + k);
                        double r = trans_b ? load_elem(B, dtype_ab, ob + n * ldb + k)
                                           : load_elem(B, dtype_ab, ob + k * ldb + n);
                        sum += l * r;
                    }
                    store_elem(C, dtype_c, oc + m * ldc + n, sum);
                } else {
                    float sum = 0.0f;
                    for (int64_t k = 0; k < K; ++k) {
                        float l = load_elem(A, dtype_ab, oa + m * lda + k);
                        float r = trans_b ? load_elem(B, dtype_ab, ob + n * ldb + k)
                                          : load_elem(B, dtype_ab, ob + k * ldb + n);
                        /* separate multiply and add, as the reference loop does (no fma
                         * contraction: the Makefile passes -ffp-contract=off) */
                        float p = l * r;
                        sum += p;
                    }
                    store_elem(C, dtype_c, oc + m * ldc + n, (double)sum);
                }
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Block-scaled GEMM (MX formats).  Restates the expectation loops of test_cmma_scaled / test_cmma_scaled_fp4
 * (crates/cubecl-core/src/runtime_tests/cmma.rs:1572-1591, :1684-1703):
 *     sum += lhs[i,l] * lhs_scale[i, l / block] * rhs[l,j] * rhs_scale[j, l / block]      (f32, left to right)
 * with A row-major [M][K], B stored [N][K] ("col-major", :1527-1529), one ue8m0 scale per `block` consecutive k
 * (block = k / scales_factor there), scales laid out [M][K/block] and [N][K/block] (:1549-1560).
 * Element types: e4m3 / e5m2 (above) and e2m1x2 = two e2m1 per byte, the FIRST element in the LOW nibble
 * (crates/cubecl-common/src/float/fp4.rs:204-216, :219-224).  e2m1 (1-2-1, bias 1; third-party `float4` 0.2.0
 * underneath, not vendored) has no NaN and no infinity; its 16 values are the OCP MX table
 * {0, 0.5, 1, 1.5, 2, 3, 4, 6} x {+,-}.  ue8m0 (fp8/fp8_e8m0.rs): value 2^(bits - 127), 0xFF = NaN.
 * ------------------------------------------------------------------------------------------ */
static const float E2M1_VALUES[8] = { 0.0f, 0.5f, 1.0f, 1.5f, 2.0f, 3.0f, 4.0f, 6.0f };
ORACLE_API float oracle_e2m1_to_f32(uint8_t nibble)
{
    const float v = E2M1_VALUES[nibble & 7u];
    return (nibble & 8u) ? -v : v;
}
/* round to nearest, ties to the even code, saturating at +-6; NaN -> +-6 is NOT pinned by the reference (e2m1 has
 * no NaN; the float4 crate's choice is unknown here) and nothing on the path produces it */
ORACLE_API uint8_t oracle_f32_to_e2m1(float f)
{
    const uint8_t sign = (f32_bits(f) >> 28) & 8u;
    const float a = fabsf(f);
    uint8_t code = 7;
    if (!(a == a)) return (uint8_t)(sign | 7u);
    for (int c = 0; c < 7; ++c) {
        const float mid = 0.5f * (E2M1_VALUES[c] + E2M1_VALUES[c + 1]);
        if (a < mid || (a == mid && (c & 1) == 0)) { code = (uint8_t)c; break; }
    }
    return (uint8_t)(sign | code);
}
ORACLE_API float oracle_ue8m0_to_f32(uint8_t b)
{
    if (b == 0xFFu) return bits_f32(0x7FC00000u);
    if (b == 0u) return bits_f32(0x00400000u);               /* 2^-127: a subnormal f32, exactly representable */
    return bits_f32((uint32_t)b << 23);
}
ORACLE_API void oracle_pack_e2m1x2(const float *src, uint8_t *dst, uint64_t n)      /* fp4.rs:204-216 */
{
    for (uint64_t i = 0; i < (n + 1) / 2; ++i) {
        const uint8_t a = oracle_f32_to_e2m1(src[2 * i]);
        const uint8_t b = (2 * i + 1 < n) ? oracle_f32_to_e2m1(src[2 * i + 1]) : 0;
        dst[i] = (uint8_t)((a & 0x0F) | ((b << 4) & 0xF0));
    }
}
ORACLE_API void oracle_unpack_e2m1x2(const uint8_t *src, float *dst, uint64_t n)
{
    for (uint64_t i = 0; i < n; ++i) dst[i] = oracle_e2m1_to_f32((uint8_t)((src[i / 2] >> ((i & 1) * 4)) & 0xF));
}

enum { DT_F4E2M1X2 = 12 };
static inline float load_mx_elem(const void *p, int dtype, int64_t idx)
{
    if (dtype == DT_F4E2M1X2) return oracle_e2m1_to_f32((uint8_t)((((const uint8_t *)p)[idx / 2] >> ((idx & 1) * 4)) & 0xF));
    return load_elem(p, dtype, idx);
}

/* lda / ldb / strides in ELEMENTS (an e2m1x2 row of K elements is K/2 bytes; lda must be even then);
 * ld_sa / ld_sb and the scale batch strides in bytes (= ue8m0 elements). */
ORACLE_API void oracle_gemm_scaled(const void *A, const uint8_t *SA, const void *B, const uint8_t *SB, void *C,
                                   int dtype_ab, int dtype_c, int64_t M, int64_t N, int64_t K, int64_t block,
                                   int64_t lda, int64_t ldb, int64_t ldc, int64_t ld_sa, int64_t ld_sb,
                                   int64_t batch, int64_t stride_a, int64_t stride_b, int64_t stride_c,
                                   int64_t stride_sa, int64_t stride_sb, int acc_f64)
{
    for (int64_t b = 0; b < batch; ++b) {
        const int64_t oa = b * stride_a, ob = b * stride_b, oc = b * stride_c;
        const uint8_t *sa = SA + b * stride_sa, *sb = SB + b * stride_sb;
        for (int64_t m = 0; m < M; ++m) {
            for (int64_t n = 0; n < N; ++n) {
                float sum = 0.0f;
                double sum64 = 0.0;
                for (int64_t k = 0; k < K; ++k) {
                    const float l = load_mx_elem(A, dtype_ab, oa + m * lda + k);
                    const float ls = oracle_ue8m0_to_f32(sa[m * ld_sa + k / block]);
                    const float r = load_mx_elem(B, dtype_ab, ob + n * ldb + k);
                    const float rs = oracle_ue8m0_to_f32(sb[n * ld_sb + k / block]);
                    if (acc_f64) {
                        sum64 += (double)l * (double)ls * (double)r * (double)rs;
                    } else {
                        float p = l * ls;          /* lhs_val * lhs_scale * rhs_val * rhs_scale, left to right */
                        p = p * r;
                        p = p * rs;
                        sum += p;
                    }
                }
                store_elem(C, dtype_c, oc + m * ldc + n, acc_f64 ? sum64 : (double)sum);
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Reductions.
 * ------------------------------------------------------------------------------------------ */

/* sum_basic (examples/sum_things/src/lib.rs:6-19): acc = 0.0f32; acc += x[i] in index order. */
ORACLE_API float oracle_sum_f32_sequential(const float *x, uint64_t n)
{
    float acc = 0.0f;
    for (uint64_t i = 0; i < n; ++i) acc += x[i];
    return acc;
}

/* f64-accumulating sum: the numerical oracle of record for array-wide sums. */
ORACLE_API double oracle_sum_f32_f64(const float *x, uint64_t n)
{
    double acc = 0.0;
    for (uint64_t i = 0; i < n; ++i) acc += (double)x[i];
    return acc;
}

ORACLE_API double oracle_sum_abs_f32_f64(const float *x, uint64_t n)
{
    double acc = 0.0;
    for (uint64_t i = 0; i < n; ++i) acc += fabs((double)x[i]);
    return acc;
}

/* Row sum over the last axis (cubecl-book .../v1-cpu.rs:7-15): out[r] = sum_j in[r*stride+j],
 * f32 accumulator starting at 0.0. */
ORACLE_API void oracle_reduce_last_axis_sum_f32(const float *in, float *out, uint64_t rows,
                                                uint64_t cols, uint64_t row_stride)
{
    for (uint64_t r = 0; r < rows; ++r) {
        float acc = 0.0f;
        for (uint64_t j = 0; j < cols; ++j) acc += in[r * row_stride + j];
        out[r] = acc;
    }
}
ORACLE_API void oracle_reduce_last_axis_sum_f64(const float *in, double *out, uint64_t rows,
                                                uint64_t cols, uint64_t row_stride)
{
    for (uint64_t r = 0; r < rows; ++r) {
        double acc = 0.0;
        for (uint64_t j = 0; j < cols; ++j) acc += (double)in[r * row_stride + j];
        out[r] = acc;
    }
}

/* Argmax ordering rule (the reference has none in-tree; SURVEY.md section 7 "hard parts"
 * defines it): the maximum under IEEE comparison, lowest index among equal maxima,
 * -0.0 == +0.0, and NaN ranks above every number with the FIRST NaN winning.  The same rule
 * is implemented on the device as an order-preserving u32 key (cubecl_amd/csrc/reduce.hip). */
static inline uint32_t argmax_key(float v)
{
    uint32_t u = f32_bits(v);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return 0xFFFFFFFFu; /* NaN: top rank          */
    if (u == 0x80000000u) u = 0;                              /* -0.0 ties with +0.0    */
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
ORACLE_API uint32_t oracle_argmax_key(float v) { return argmax_key(v); }

/* returns the index; *out_val receives x[index] (bit-exact copy).  n == 0 -> index 0 and
 * *out_val = -inf (the identity), mirroring the device path. */
ORACLE_API uint64_t oracle_argmax_f32(const float *x, uint64_t n, float *out_val)
{
    if (n == 0) { if (out_val) *out_val = -INFINITY; return 0; }
    uint64_t best = 0;
    uint32_t best_key = argmax_key(x[0]);
    for (uint64_t i = 1; i < n; ++i) {
        uint32_t k = argmax_key(x[i]);
        if (k > best_key) { best_key = k; best = i; }
    }
    if (out_val) *out_val = x[best];
    return best;
}

/* Argmin: the mirror image (review of round 3, missing #3) -- the minimum under IEEE comparison, lowest index among equal
 * minima, -0.0 == +0.0, NaN ranks FIRST with the first NaN winning (numpy's argmin).  The device mirrors the key (~key) and
 * keeps the NaN code on top (cubecl_amd/csrc/reduce.hip arg_key<AOP_MIN>); restated here without the key trick.
 * n == 0 -> index 0 and +inf. */
ORACLE_API uint64_t oracle_argmin_f32(const float *x, uint64_t n, float *out_val)
{
    if (n == 0) { if (out_val) *out_val = INFINITY; return 0; }
    uint64_t best = 0;
    int best_nan = x[0] != x[0];
    for (uint64_t i = 1; i < n && !best_nan; ++i) {
        if (x[i] != x[i]) { best = i; best_nan = 1; }
        else if (x[i] < x[best]) best = i;               /* strict: equal values (and -0 vs +0) keep the lower index */
    }
    if (out_val) *out_val = x[best];
    return best;
}

/* Value reductions max / min (mi355_reduce, mi355_reduce_axis): the IEEE maximum / minimum with -0 < +0, NaN if ANY element is
 * NaN (numpy's max / min); empty input gives the identity -inf / +inf.  Restates plane_max / plane_min
 * (crates/cubecl-core/src/frontend/plane.rs:352, :370) over a whole array with the NaN rule made explicit (the plane forms
 * lower to max(a, b) / min(a, b), whose NaN behaviour the reference does not pin). */
ORACLE_API float oracle_max_f32(const float *x, uint64_t n)
{
    float m = -INFINITY;
    for (uint64_t i = 0; i < n; ++i) {
        if (x[i] != x[i]) return bits_f32(0x7FC00000u);
        if (x[i] > m || (x[i] == m && f32_bits(m) == 0x80000000u)) m = x[i];      /* +0 beats -0 */
    }
    return m;
}
ORACLE_API float oracle_min_f32(const float *x, uint64_t n)
{
    float m = INFINITY;
    for (uint64_t i = 0; i < n; ++i) {
        if (x[i] != x[i]) return bits_f32(0x7FC00000u);
        if (x[i] < m || (x[i] == m && f32_bits(x[i]) == 0x80000000u)) m = x[i];   /* -0 beats +0 */
    }
    return m;
}
/* product in f64 (the numerical oracle of MI355_REDUCE_PROD: the device multiplies in f32 in its summation tree's shape) */
ORACLE_API double oracle_prod_f32_f64(const float *x, uint64_t n)
{
    double p = 1.0;
    for (uint64_t i = 0; i < n; ++i) p *= (double)x[i];
    return p;
}

ORACLE_API void oracle_reduce_last_axis_argmax_f32(const float *in, uint32_t *out_idx,
                                                   uint64_t rows, uint64_t cols,
                                                   uint64_t row_stride)
{
    for (uint64_t r = 0; r < rows; ++r)
        out_idx[r] = (uint32_t)oracle_argmax_f32(in + r * row_stride, cols, NULL);
}

/* plane_reduce (crates/cubecl-cpp/src/shared/plane.rs:60-70): xor butterfly, offsets
 * 1,2,4,... < width; every lane ends with the result.  vals has `width` lanes (power of two);
 * op: 0 sum, 1 prod, 2 max, 3 min.  In-place. */
ORACLE_API void oracle_plane_reduce_f32(float *vals, uint32_t width, int op)
{
    float tmp[1024];
    if (width > 1024) return;
    for (uint32_t off = 1; off < width; off *= 2) {
        for (uint32_t l = 0; l < width; ++l) {
            float a = vals[l], b = vals[l ^ off];
            switch (op) {
            case 0: tmp[l] = a + b; break;
            case 1: tmp[l] = a * b; break;
            case 2: tmp[l] = a > b ? a : b; break;
            default: tmp[l] = a < b ? a : b; break;
            }
        }
        memcpy(vals, tmp, width * sizeof(float));
    }
}

/* plane_reduce_inclusive (shared/plane.rs:72-88): Hillis-Steele scan with shuffle_up. */
ORACLE_API void oracle_plane_inclusive_sum_f32(float *vals, uint32_t width)
{
    float tmp[1024];
    if (width > 1024) return;
    for (uint32_t off = 1; off < width; off *= 2) {
        for (uint32_t l = 0; l < width; ++l)
            tmp[l] = (l >= off) ? vals[l] + vals[l - off] : vals[l];
        memcpy(vals, tmp, width * sizeof(float));
    }
}
/* plane_inclusive_sum / _prod and plane_exclusive_sum / _prod (crates/cubecl-core/src/frontend/plane.rs:242-283, :309, :334) as
 * the HIP backend lowers them: plane_reduce_inclusive with OpAdd / OpMul, and plane_reduce_exclusive = the inclusive scan
 * shuffled up by one lane with lane 0 taking the default 0 / 1 (crates/cubecl-cpp/src/shared/plane.rs:72-97, :128-136). */
ORACLE_API void oracle_plane_scan_f32(float *vals, uint32_t width, int mul, int exclusive)
{
    float tmp[1024];
    if (width > 1024 || width == 0) return;
    for (uint32_t off = 1; off < width; off *= 2) {
        for (uint32_t l = 0; l < width; ++l) {
            const float up = (l >= off) ? vals[l - off] : vals[l];
            tmp[l] = (l >= off) ? (mul ? vals[l] * up : vals[l] + up) : vals[l];
        }
        memcpy(vals, tmp, width * sizeof(float));
    }
    if (exclusive) {
        for (uint32_t l = width - 1; l > 0; --l) vals[l] = vals[l - 1];
        vals[0] = mul ? 1.0f : 0.0f;
    }
}

/* ------------------------------------------------------------------------------------------
 * cubecl-cpu execution-model restatement for the timed CPU baseline
 * (crates/cubecl-cpu/src/compute/threadpool/mod.rs:80-99: one worker per cube UNIT, the cube
 * count is a loop inside each unit; crates/cubecl-std/src/throughput/base.rs:186-197:
 * `num_cpu_cores` units).  Each unit reduces a contiguous slice sequentially in f32; the
 * unit partials are added in unit order.  Not the numerical oracle -- the CPU baseline.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    const float *x; uint64_t begin, end; float sum; uint32_t key; uint64_t idx;
} reduce_job;

static void *reduce_worker(void *arg)
{
    reduce_job *j = (reduce_job *)arg;
    float acc = 0.0f;
    uint32_t bk = 0; uint64_t bi = j->begin; int have = 0;
    for (uint64_t i = j->begin; i < j->end; ++i) {
        float v = j->x[i];
        acc += v;
        uint32_t k = argmax_key(v);
        if (!have || k > bk) { bk = k; bi = i; have = 1; }
    }
    j->sum = acc; j->key = bk; j->idx = bi;
    return NULL;
}

/* fused sum + argmax with `units` worker threads; returns seconds of wall time. */
ORACLE_API double oracle_cpu_sum_argmax_f32(const float *x, uint64_t n, int units,
                                            float *out_sum, uint64_t *out_idx)
{
    if (units < 1) units = 1;
    if (units > 1024) units = 1024;
    pthread_t th[1024];
    reduce_job jobs[1024];
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    uint64_t per = (n + (uint64_t)units - 1) / (uint64_t)units;
    for (int u = 0; u < units; ++u) {
        uint64_t b = per * (uint64_t)u, e = b + per;
        if (b > n) b = n;
        if (e > n) e = n;
        jobs[u].x = x; jobs[u].begin = b; jobs[u].end = e;
        pthread_create(&th[u], NULL, reduce_worker, &jobs[u]);
    }
    float sum = 0.0f; uint32_t bk = 0; uint64_t bi = 0; int have = 0;
    for (int u = 0; u < units; ++u) {
        pthread_join(th[u], NULL);
        sum += jobs[u].sum;
        if (jobs[u].end > jobs[u].begin && (!have || jobs[u].key > bk)) {
            bk = jobs[u].key; bi = jobs[u].idx; have = 1;
        }
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (out_sum) *out_sum = sum;
    if (out_idx) *out_idx = bi;
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

typedef struct {
    const void *A, *B; void *C; int dtype_ab, dtype_c;
    int64_t m0, m1, N, K, lda, ldb, ldc; int trans_b;
} gemm_job;

/* Blocked over n (64-wide strips held in f32 accumulators) but k stays the innermost
 * sequential f32 accumulation per output, i.e. the same arithmetic as oracle_gemm. Inputs are
 * widened once per row/strip to keep the baseline from being conversion-bound. */
static void *gemm_worker(void *arg)
{
    gemm_job *j = (gemm_job *)arg;
    const int64_t N = j->N, K = j->K;
    float *arow = (float *)malloc((size_t)K * sizeof(float));
    float *bcol = (float *)malloc((size_t)K * 64 * sizeof(float));
    for (int64_t n0 = 0; n0 < N; n0 += 64) {
        const int64_t nb = (N - n0 < 64) ? (N - n0) : 64;
        for (int64_t k = 0; k < K; ++k)
            for (int64_t n = 0; n < nb; ++n)
                bcol[k * 64 + n] = j->trans_b ? load_elem(j->B, j->dtype_ab, (n0 + n) * j->ldb + k)
                                              : load_elem(j->B, j->dtype_ab, k * j->ldb + n0 + n);
        for (int64_t m = j->m0; m < j->m1; ++m) {
            for (int64_t k = 0; k < K; ++k) arow[k] = load_elem(j->A, j->dtype_ab, m * j->lda + k);
            float acc[64];
            for (int64_t n = 0; n < 64; ++n) acc[n] = 0.0f;
            for (int64_t k = 0; k < K; ++k) {
                const float a = arow[k];
                const float *bk = bcol + k * 64;
                for (int64_t n = 0; n < 64; ++n) { float p = a * bk[n]; acc[n] += p; }
            }
            for (int64_t n = 0; n < nb; ++n)
                store_elem(j->C, j->dtype_c, m * j->ldc + n0 + n, (double)acc[n]);
        }
    }
    free(arow); free(bcol);
    return NULL;
}

/* Threaded CPU GEMM (single batch); returns seconds of wall time. */
ORACLE_API double oracle_cpu_gemm(const void *A, const void *B, void *C, int dtype_ab,
                                  int dtype_c, int64_t M, int64_t N, int64_t K, int64_t lda,
                                  int64_t ldb, int64_t ldc, int trans_b, int units)
{
    if (units < 1) units = 1;
    if (units > 1024) units = 1024;
    pthread_t th[1024];
    gemm_job jobs[1024];
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    int64_t per = (M + units - 1) / units;
    for (int u = 0; u < units; ++u) {
        int64_t b = per * u, e = b + per;
        if (b > M) b = M;
        if (e > M) e = M;
        gemm_job jb = { A, B, C, dtype_ab, dtype_c, b, e, N, K, lda, ldb, ldc, trans_b };
        jobs[u] = jb;
        pthread_create(&th[u], NULL, gemm_worker, &jobs[u]);
    }
    for (int u = 0; u < units; ++u) pthread_join(th[u], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

ORACLE_API int oracle_abi_version(void) { return 1; }
