"""ctypes + numpy front-end of oracle/liboracle.so (the C restatement of the reference path).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg -- never by cubecl_amd (the product path).
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_DIR = Path(__file__).resolve().parent
LIB_PATH = _DIR / "liboracle.so"
SEED = 0x5EEDC0BE
DT_F32, DT_BF16, DT_F16 = 0, 1, 2
DT_F8E4M3, DT_F8E5M2 = 10, 11
DT_F4E2M1X2, DT_UE8M0 = 12, 13

_lib = None


def build() -> None:
    subprocess.run(["make", "-C", str(_DIR), "-s"], check=True)


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        build()
    L = C.CDLL(str(LIB_PATH))
    P, u64, i64, i32, f32 = C.c_void_p, C.c_uint64, C.c_int64, C.c_int, C.c_float
    L.oracle_fill_uniform_f32.argtypes = [P, u64, u64, u64, f32, f32]
    L.oracle_fill_uniform_f32.restype = None
    L.oracle_fill_uniform_f32_at.argtypes = [P, u64, u64, u64, u64, f32, f32]
    L.oracle_fill_uniform_f32_at.restype = None
    for name in ("oracle_convert_f32_to_bf16", "oracle_convert_f32_to_f16", "oracle_convert_bf16_to_f32",
                 "oracle_convert_f16_to_f32"):
        getattr(L, name).argtypes = [P, P, u64]
        getattr(L, name).restype = None
    L.oracle_gemm.argtypes = [P, P, P, i32, i32, i64, i64, i64, i64, i64, i64, i32, i64, i64, i64, i64, i32]
    L.oracle_gemm.restype = None
    L.oracle_sum_f32_sequential.argtypes = [P, u64]
    L.oracle_sum_f32_sequential.restype = f32
    L.oracle_sum_f32_f64.argtypes = [P, u64]
    L.oracle_sum_f32_f64.restype = C.c_double
    L.oracle_sum_abs_f32_f64.argtypes = [P, u64]
    L.oracle_sum_abs_f32_f64.restype = C.c_double
    L.oracle_reduce_last_axis_sum_f32.argtypes = [P, P, u64, u64, u64]
    L.oracle_reduce_last_axis_sum_f32.restype = None
    L.oracle_reduce_last_axis_sum_f64.argtypes = [P, P, u64, u64, u64]
    L.oracle_reduce_last_axis_sum_f64.restype = None
    L.oracle_argmax_f32.argtypes = [P, u64, C.POINTER(f32)]
    L.oracle_argmax_f32.restype = u64
    L.oracle_argmin_f32.argtypes = [P, u64, C.POINTER(f32)]
    L.oracle_argmin_f32.restype = u64
    L.oracle_max_f32.argtypes = [P, u64]
    L.oracle_max_f32.restype = f32
    L.oracle_min_f32.argtypes = [P, u64]
    L.oracle_min_f32.restype = f32
    L.oracle_prod_f32_f64.argtypes = [P, u64]
    L.oracle_prod_f32_f64.restype = C.c_double
    L.oracle_plane_scan_f32.argtypes = [P, C.c_uint32, i32, i32]
    L.oracle_argmax_key.argtypes = [f32]
    L.oracle_argmax_key.restype = C.c_uint32
    L.oracle_reduce_last_axis_argmax_f32.argtypes = [P, P, u64, u64, u64]
    L.oracle_reduce_last_axis_argmax_f32.restype = None
    L.oracle_plane_reduce_f32.argtypes = [P, C.c_uint32, i32]
    L.oracle_plane_reduce_f32.restype = None
    L.oracle_plane_inclusive_sum_f32.argtypes = [P, C.c_uint32]
    L.oracle_plane_inclusive_sum_f32.restype = None
    L.oracle_cpu_sum_argmax_f32.argtypes = [P, u64, i32, C.POINTER(f32), C.POINTER(u64)]
    L.oracle_cpu_sum_argmax_f32.restype = C.c_double
    L.oracle_convert_f32_to_fp8.argtypes = [P, P, u64, i32]
    L.oracle_convert_f32_to_fp8.restype = None
    L.oracle_convert_fp8_to_f32.argtypes = [P, P, u64, i32]
    L.oracle_convert_fp8_to_f32.restype = None
    L.oracle_pack_e2m1x2.argtypes = [P, P, u64]
    L.oracle_pack_e2m1x2.restype = None
    L.oracle_unpack_e2m1x2.argtypes = [P, P, u64]
    L.oracle_unpack_e2m1x2.restype = None
    L.oracle_ue8m0_to_f32.argtypes = [C.c_uint8]
    L.oracle_ue8m0_to_f32.restype = f32
    L.oracle_gemm_scaled.argtypes = [P, P, P, P, P, i32, i32] + [i64] * 15 + [i32]
    L.oracle_gemm_scaled.restype = None
    L.oracle_cpu_gemm.argtypes = [P, P, P, i32, i32, i64, i64, i64, i64, i64, i64, i32, i32]
    L.oracle_cpu_gemm.restype = C.c_double
    _lib = L
    return L


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def fill_uniform(n: int, tensor: int, lo: float, hi: float, seed: int = SEED) -> np.ndarray:
    out = np.empty(n, dtype=np.float32)
    lib().oracle_fill_uniform_f32(_p(out), n, seed, tensor, lo, hi)
    return out


def fill_uniform_at(start: int, n: int, tensor: int, lo: float, hi: float, seed: int = SEED) -> np.ndarray:
    """Elements [start, start + n) of fill_uniform's stream (counter RNG: no need to generate what lies in front)."""
    out = np.empty(n, dtype=np.float32)
    lib().oracle_fill_uniform_f32_at(_p(out), start, n, seed, tensor, lo, hi)
    return out


def to_bf16(x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty(x.shape, dtype=np.uint16)
    lib().oracle_convert_f32_to_bf16(_p(x), _p(out), x.size)
    return out


def to_f16(x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty(x.shape, dtype=np.uint16)
    lib().oracle_convert_f32_to_f16(_p(x), _p(out), x.size)
    return out


def from_bf16(x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.uint16)
    out = np.empty(x.shape, dtype=np.float32)
    lib().oracle_convert_bf16_to_f32(_p(x), _p(out), x.size)
    return out


def from_f16(x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.uint16)
    out = np.empty(x.shape, dtype=np.float32)
    lib().oracle_convert_f16_to_f32(_p(x), _p(out), x.size)
    return out


def to_fp8(x: np.ndarray, dtype: int = DT_F8E4M3) -> np.ndarray:
    """f32 -> OCP FP8 bits (RNE, saturating to +-MAX, NaN kept): e4m3::from_f32 / e5m2::from_f32
    (crates/cubecl-common/src/float/fp8/fp8_e4m3.rs:77-87, fp8_e5m2.rs:78-88)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty(x.shape, dtype=np.uint8)
    lib().oracle_convert_f32_to_fp8(_p(x), _p(out), x.size, int(dtype == DT_F8E5M2))
    return out


def from_fp8(x: np.ndarray, dtype: int = DT_F8E4M3) -> np.ndarray:
    """OCP FP8 bits -> f32, exact (fp8_e4m3.rs:110-115)."""
    x = np.ascontiguousarray(x, dtype=np.uint8)
    out = np.empty(x.shape, dtype=np.float32)
    lib().oracle_convert_fp8_to_f32(_p(x), _p(out), x.size, int(dtype == DT_F8E5M2))
    return out


def pack_e2m1x2(x: np.ndarray) -> np.ndarray:
    """f32 -> packed e2m1 pairs, first element in the low nibble (e2m1x2::from_f32_slice, fp4.rs:204-216)."""
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1)
    out = np.empty((x.size + 1) // 2, dtype=np.uint8)
    lib().oracle_pack_e2m1x2(_p(x), _p(out), x.size)
    return out


def unpack_e2m1x2(bits: np.ndarray, n: int | None = None) -> np.ndarray:
    bits = np.ascontiguousarray(bits, dtype=np.uint8).reshape(-1)
    n = 2 * bits.size if n is None else n
    out = np.empty(n, dtype=np.float32)
    lib().oracle_unpack_e2m1x2(_p(bits), _p(out), n)
    return out


def from_ue8m0(bits: np.ndarray) -> np.ndarray:
    """ue8m0 -> f32: 2^(bits - 127), 0xFF = NaN (fp8/fp8_e8m0.rs)."""
    bits = np.ascontiguousarray(bits, dtype=np.uint8)
    return np.array([lib().oracle_ue8m0_to_f32(int(b)) for b in bits.reshape(-1)], dtype=np.float32).reshape(bits.shape)


def gemm_scaled(a, sa, b, sb, m: int, n: int, k: int, *, dtype_ab: int, dtype_c: int = DT_F32, block: int = 32,
                lda=None, ldb=None, ldc=None, ld_sa=None, ld_sb=None, batch: int = 1, stride_a=None, stride_b=None,
                stride_c=None, stride_sa=None, stride_sb=None, acc_f64: bool = False) -> np.ndarray:
    """Block-scaled C = (A .* SA) (B .* SB)^T following test_cmma_scaled (runtime_tests/cmma.rs:1572-1591)."""
    a = np.ascontiguousarray(a, dtype=np.uint8).reshape(-1)
    b = np.ascontiguousarray(b, dtype=np.uint8).reshape(-1)
    sa = np.ascontiguousarray(sa, dtype=np.uint8).reshape(-1)
    sb = np.ascontiguousarray(sb, dtype=np.uint8).reshape(-1)
    lda = k if lda is None else lda
    ldb = k if ldb is None else ldb
    ldc = n if ldc is None else ldc
    ld_sa = k // block if ld_sa is None else ld_sa
    ld_sb = k // block if ld_sb is None else ld_sb
    stride_a = m * lda if stride_a is None else stride_a
    stride_b = n * ldb if stride_b is None else stride_b
    stride_c = m * ldc if stride_c is None else stride_c
    stride_sa = m * ld_sa if stride_sa is None else stride_sa
    stride_sb = n * ld_sb if stride_sb is None else stride_sb
    c = np.zeros(max(batch * stride_c, m * ldc), dtype=_NP_OF[dtype_c])
    lib().oracle_gemm_scaled(_p(a), _p(sa), _p(b), _p(sb), _p(c), dtype_ab, dtype_c, m, n, k, block, lda, ldb, ldc, ld_sa, ld_sb,
                             batch, stride_a, stride_b, stride_c, stride_sa, stride_sb, int(acc_f64))
    return c


_NP_OF = {DT_F32: np.float32, DT_BF16: np.uint16, DT_F16: np.uint16, DT_F8E4M3: np.uint8, DT_F8E5M2: np.uint8}


def gemm(a: np.ndarray, b: np.ndarray, m: int, n: int, k: int, *, dtype_ab: int = DT_F32, dtype_c: int = DT_F32,
         lda: int | None = None, ldb: int | None = None, ldc: int | None = None, trans_b: bool = False,
         batch: int = 1, stride_a: int | None = None, stride_b: int | None = None, stride_c: int | None = None,
         acc_f64: bool = False) -> np.ndarray:
    """C = A * B (or A * B^T with trans_b) following runtime_tests/cmma.rs:695-722."""
    a = np.ascontiguousarray(a, dtype=_NP_OF[dtype_ab]).reshape(-1)
    b = np.ascontiguousarray(b, dtype=_NP_OF[dtype_ab]).reshape(-1)
    lda = k if lda is None else lda
    ldb = (k if trans_b else n) if ldb is None else ldb
    ldc = n if ldc is None else ldc
    stride_a = m * lda if stride_a is None else stride_a
    stride_b = ((n if trans_b else k) * ldb) if stride_b is None else stride_b
    stride_c = m * ldc if stride_c is None else stride_c
    c = np.zeros(max(batch * stride_c, m * ldc), dtype=_NP_OF[dtype_c])
    lib().oracle_gemm(_p(a), _p(b), _p(c), dtype_ab, dtype_c, m, n, k, lda, ldb, ldc, int(trans_b), batch,
                      stride_a, stride_b, stride_c, int(acc_f64))
    return c


def gemm_add(a, b, c, m: int, n: int, k: int, *, dtype_ab: int = DT_F32, dtype_c: int = DT_F32, trans_b: bool = False,
             batch: int = 1) -> np.ndarray:
    """D = A * B + C, cmma::execute(a, b, c, d) (crates/cubecl-core/src/frontend/cmma.rs:1066-1110): the accumulator
    fragment is f32, so the product (oracle_gemm's sequential f32 sum, runtime_tests/cmma.rs:695-722) and C meet in f32 and
    the result is rounded once to dtype_c.  Contiguous operands; c holds dtype_c values (bit patterns for 16-bit)."""
    prod = gemm(a, b, m, n, k, dtype_ab=dtype_ab, dtype_c=DT_F32, trans_b=trans_b, batch=batch)
    c = np.asarray(c).reshape(-1)
    cw = c.astype(np.float32) if dtype_c == DT_F32 else (from_bf16(c) if dtype_c == DT_BF16 else from_f16(c))
    d = (prod.reshape(-1) + cw).astype(np.float32)
    if dtype_c == DT_F32:
        return d
    return to_bf16(d) if dtype_c == DT_BF16 else to_f16(d)


def sum_sequential(x: np.ndarray) -> float:
    x = np.ascontiguousarray(x, dtype=np.float32)
    return float(lib().oracle_sum_f32_sequential(_p(x), x.size))


def sum_f64(x: np.ndarray) -> float:
    x = np.ascontiguousarray(x, dtype=np.float32)
    return float(lib().oracle_sum_f32_f64(_p(x), x.size))


def sum_abs_f64(x: np.ndarray) -> float:
    x = np.ascontiguousarray(x, dtype=np.float32)
    return float(lib().oracle_sum_abs_f32_f64(_p(x), x.size))


def argmax(x: np.ndarray):
    x = np.ascontiguousarray(x, dtype=np.float32)
    val = C.c_float()
    idx = lib().oracle_argmax_f32(_p(x), x.size, C.byref(val))
    return int(idx), np.float32(val.value)


def argmin(x: np.ndarray):
    """(index, value) of the minimum: lowest index wins ties, -0 == +0, the first NaN wins (oracle_argmin_f32)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    val = C.c_float()
    idx = lib().oracle_argmin_f32(_p(x), x.size, C.byref(val))
    return int(idx), np.float32(val.value)


def reduce_value(x: np.ndarray, op: str):
    """Array-wide value reduction of mi355_reduce: "sum" / "mean" / "prod" in f64 (numerical oracle of the f32 tree), "max" /
    "min" exact (NaN if any NaN, -0 < +0, empty: -inf / +inf)."""
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1)
    if op == "sum":
        return float(lib().oracle_sum_f32_f64(_p(x), x.size))
    if op == "mean":
        return float(lib().oracle_sum_f32_f64(_p(x), x.size)) / float(np.float32(x.size)) if x.size else 0.0
    if op == "prod":
        return float(lib().oracle_prod_f32_f64(_p(x), x.size))
    if op == "max":
        return np.float32(lib().oracle_max_f32(_p(x), x.size))
    if op == "min":
        return np.float32(lib().oracle_min_f32(_p(x), x.size))
    raise ValueError(op)


def reduce_axis_value(x: np.ndarray, axis: int, op: str) -> np.ndarray:
    """The same along one axis (rows of the moved axis through reduce_value; f64 for sum / mean / prod)."""
    x = np.asarray(x, dtype=np.float32)
    moved = np.ascontiguousarray(np.moveaxis(x, axis, -1))
    rows = moved.reshape(-1, moved.shape[-1]) if moved.shape[-1] else moved.reshape(-1, 0)
    out = np.array([reduce_value(r, op) for r in rows], dtype=np.float32 if op in ("max", "min") else np.float64)
    if op == "mean" and moved.shape[-1] == 0:
        out[:] = np.nan
    return out.reshape(moved.shape[:-1])


def reduce_axis_argmin(x: np.ndarray, axis: int) -> np.ndarray:
    x = np.asarray(x, dtype=np.float32)
    moved = np.ascontiguousarray(np.moveaxis(x, axis, -1))
    rows = moved.reshape(-1, moved.shape[-1])
    return np.array([argmin(r)[0] for r in rows], dtype=np.uint32).reshape(moved.shape[:-1])


def plane_scan(vals: np.ndarray, mul: bool, exclusive: bool) -> np.ndarray:
    """plane_inclusive / exclusive sum / prod over ONE plane of len(vals) lanes (oracle_plane_scan_f32)."""
    v = np.ascontiguousarray(vals, dtype=np.float32).copy()
    lib().oracle_plane_scan_f32(_p(v), v.size, int(mul), int(exclusive))
    return v


def reduce_last_axis_sum(x: np.ndarray, f64: bool = False) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    rows, cols = (int(np.prod(x.shape[:-1])), x.shape[-1])
    if f64:
        out = np.empty(rows, dtype=np.float64)
        lib().oracle_reduce_last_axis_sum_f64(_p(x), _p(out), rows, cols, cols)
    else:
        out = np.empty(rows, dtype=np.float32)
        lib().oracle_reduce_last_axis_sum_f32(_p(x), _p(out), rows, cols, cols)
    return out.reshape(x.shape[:-1])


def reduce_last_axis_argmax(x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    rows, cols = (int(np.prod(x.shape[:-1])), x.shape[-1])
    out = np.empty(rows, dtype=np.uint32)
    lib().oracle_reduce_last_axis_argmax_f32(_p(x), _p(out), rows, cols, cols)
    return out.reshape(x.shape[:-1])


def reduce_axis_sum(x: np.ndarray, axis: int) -> np.ndarray:
    """f64 sum over one axis (numpy restatement of the per-unit loop of the book's reduce kernels along `axis`)."""
    return np.asarray(x, dtype=np.float32).astype(np.float64).sum(axis=axis)


def reduce_axis_argmax(x: np.ndarray, axis: int) -> np.ndarray:
    """Lowest index of the maximum along `axis` with the argmax_key order (NaN highest, -0 == +0)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    moved = np.ascontiguousarray(np.moveaxis(x, axis, -1))
    return reduce_last_axis_argmax(moved)


def plane_reduce(vals: np.ndarray, op: int) -> np.ndarray:
    v = np.ascontiguousarray(vals, dtype=np.float32).copy()
    lib().oracle_plane_reduce_f32(_p(v), v.size, op)
    return v


def plane_inclusive_sum(vals: np.ndarray) -> np.ndarray:
    v = np.ascontiguousarray(vals, dtype=np.float32).copy()
    lib().oracle_plane_inclusive_sum_f32(_p(v), v.size)
    return v


PLANE_ALL, PLANE_ANY, PLANE_ELECT, PLANE_BROADCAST, PLANE_SHUFFLE, PLANE_SHUFFLE_XOR, PLANE_SHUFFLE_UP, PLANE_SHUFFLE_DOWN, PLANE_BALLOT = range(200, 209)


def plane_op(vals: np.ndarray, op: int, plane: int, arg: int = 0) -> np.ndarray:
    """The remaining plane intrinsics (crates/cubecl-core/src/frontend/plane.rs:62-216, :388-440) over planes of `plane` lanes
    -- numpy restatement of the semantics the reference's tests pin (runtime_tests/plane.rs:527-850) and its HIP lowering gives
    (crates/cubecl-cpp/src/hip/plane.rs:19-64: a shuffle whose source lies outside the plane keeps the lane's own value;
    shared/plane.rs:170-174: elect = the lowest active lane).  ALL / ANY test "non-zero"; BALLOT returns 4 x u32 per plane."""
    v = np.ascontiguousarray(vals, dtype=np.float32).reshape(-1)
    out = np.empty((-(-v.size // plane), 4), dtype=np.uint32) if op == PLANE_BALLOT else v.copy()
    for p0 in range(0, v.size, plane):
        x = v[p0:p0 + plane]
        lanes = np.arange(x.size)
        if op == PLANE_ALL:
            out[p0:p0 + plane] = 1.0 if np.all(x != 0) else 0.0
        elif op == PLANE_ANY:
            out[p0:p0 + plane] = 1.0 if np.any(x != 0) else 0.0
        elif op == PLANE_ELECT:
            out[p0:p0 + plane] = (lanes == 0).astype(np.float32)
        elif op in (PLANE_BROADCAST, PLANE_SHUFFLE):
            out[p0:p0 + plane] = x[arg] if arg < x.size else x
        elif op == PLANE_SHUFFLE_XOR:
            src = lanes ^ arg
            out[p0:p0 + plane] = np.where(src < x.size, x[np.minimum(src, x.size - 1)], x)
        elif op == PLANE_SHUFFLE_UP:
            src = lanes - arg
            out[p0:p0 + plane] = np.where(src >= 0, x[np.maximum(src, 0)], x)
        elif op == PLANE_SHUFFLE_DOWN:
            src = lanes + arg
            out[p0:p0 + plane] = np.where(src < x.size, x[np.minimum(src, x.size - 1)], x)
        elif op == PLANE_BALLOT:
            m = int(sum(1 << int(i) for i in lanes[x != 0]))
            out[p0 // plane] = [m & 0xFFFFFFFF, (m >> 32) & 0xFFFFFFFF, 0, 0]
        else:
            raise ValueError(op)
    return out


def cpu_sum_argmax(x: np.ndarray, units: int):
    x = np.ascontiguousarray(x, dtype=np.float32)
    s, i = C.c_float(), C.c_uint64()
    secs = lib().oracle_cpu_sum_argmax_f32(_p(x), x.size, units, C.byref(s), C.byref(i))
    return secs, float(s.value), int(i.value)


def cpu_gemm(a, b, m, n, k, *, dtype_ab=DT_F32, dtype_c=DT_F32, trans_b=False, units=1):
    a = np.ascontiguousarray(a, dtype=_NP_OF[dtype_ab]).reshape(-1)
    b = np.ascontiguousarray(b, dtype=_NP_OF[dtype_ab]).reshape(-1)
    c = np.zeros(m * n, dtype=_NP_OF[dtype_c])
    secs = lib().oracle_cpu_gemm(_p(a), _p(b), _p(c), dtype_ab, dtype_c, m, n, k, k, (k if trans_b else n), n,
                                 int(trans_b), units)
    return secs, c
