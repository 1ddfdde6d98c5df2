"""Parity of the HIP GEMM kernels with the CPU oracle through the C ABI.

Tolerances (written here, per BASELINE.json "within 1e-5 relative for f32"):
  * integer-valued known-answer vectors of the reference: exact (assert_eq! in the reference);
  * f32 outputs: |C - C_ref| <= 1e-5 * (|A| |B|)_ij, C_ref accumulated in f64 from the same
    (already rounded) inputs -- MFMA products are exact in f32, only the summation order differs;
  * 16-bit outputs: within one unit in the last place of the 16-bit format;
  * fp8 operands on the MFMA kernels: plus what the instruction itself drops (fp8_mfma_truncation_bound below: the hardware is
    not an exact-product f32 accumulation; bf16 / f16 / f32 MFMA are, to 1e-7 of sum |a||b| at 8192^3 -- profiles/r03_parity_margins.jsonl).
"""
import ctypes as C
import json
from pathlib import Path

import os

import numpy as np
import pytest

from cubecl_amd import ElemType, ServerError, TensorHandle, ops
from cubecl_amd import _native as N

pytestmark = pytest.mark.gpu
GOLD = json.loads((Path(__file__).parent / "golden" / "reference_goldens.json").read_text())
REL = 1e-5
ALGOS = {"auto": N.GEMM_ALGO_AUTO, "generic": N.GEMM_ALGO_GENERIC, "f32": N.GEMM_ALGO_F32_MFMA,
         "lp128": N.GEMM_ALGO_LP_128, "lp256": N.GEMM_ALGO_LP_256, "lp256w4": N.GEMM_ALGO_LP_256W4, "lp256p": N.GEMM_ALGO_LP_256P,
         "lp256q": N.GEMM_ALGO_LP_256Q, "skinny": N.GEMM_ALGO_SKINNY, "stream64": N.GEMM_ALGO_STREAM64,
         "lp256x128": N.GEMM_ALGO_LP_256X128, "nnrows": N.GEMM_ALGO_NNROWS, "lp256x192": N.GEMM_ALGO_LP_256X192, "lp192x192": N.GEMM_ALGO_LP_192X192, "lp256m16": N.GEMM_ALGO_LP_256M16, "lp256qm": N.GEMM_ALGO_LP_256QM}


def _to_dev(client, oracle, x, dtype):
    if dtype == ElemType.F32:
        return TensorHandle.from_numpy(client, x.astype(np.float32)), x.astype(np.float32)
    if dtype in (ElemType.F8E4M3, ElemType.F8E5M2):
        bits = oracle.to_fp8(x, int(dtype))
        return TensorHandle.from_numpy(client, bits, dtype), oracle.from_fp8(bits, int(dtype))
    bits = oracle.to_bf16(x) if dtype == ElemType.BF16 else oracle.to_f16(x)
    back = oracle.from_bf16(bits) if dtype == ElemType.BF16 else oracle.from_f16(bits)
    return TensorHandle.from_numpy(client, bits, dtype), back


def _decode(oracle, arr, dtype):
    if dtype == ElemType.F32:
        return arr.astype(np.float64)
    return (oracle.from_bf16(arr) if dtype == ElemType.BF16 else oracle.from_f16(arr.view(np.uint16))).astype(np.float64)


def fp8_mfma_truncation_bound(A, Bt):
    """What `v_mfma_f32_32x32x64_f8f6f4` may lose against exact products accumulated in f32 (measured on gfx950,
    tools/dev/fp8_mfma_exactness.py -> profiles/r03_fp8_mfma_internal_width.txt): the instruction adds its products in groups of 8
    consecutive k, each group aligned to its largest product and cut 13 bits below that product's exponent -- 1 + 63 x 2^-14 comes
    out as 1 + 56 x 2^-14 (the 7 small products that share a group with the 1 are gone, the other 56 are all there), and 2^-13 still
    survives.  So every product but the largest of its group loses less than 2^-13 of that largest one:
    |err| < 7 * 2^-13 * sum over groups of max_k |a_k b_k| <= 7 * 2^-13 * (groupmax|A| @ groupmax|B|).  A [m][k], Bt [k][n] in f64;
    groups start at multiples of 8 (every kernel starts its K walk, and every split-K slice, on a multiple of 128)."""
    m, k = A.shape
    kp = (k + 7) // 8 * 8
    a8 = np.zeros((m, kp)); a8[:, :k] = np.abs(A)
    b8 = np.zeros((kp, Bt.shape[1])); b8[:k] = np.abs(Bt)
    return 7.0 * 2.0 ** -13 * (a8.reshape(m, kp // 8, 8).max(axis=2) @ b8.reshape(kp // 8, 8, -1).max(axis=1))


def run_case(client, oracle, m, n, k, dtype, out_dtype, trans_b, algo, *, lda=None, ldb=None, ldc=None, batch=1,
             bcast_b=False, seed_t=50, trans_a=False):
    """Random [-1,1) operands; returns nothing, asserts parity.  trans_a: A stored [K][M] (lda >= M)."""
    lda = lda or (m if trans_a else k)
    ldb = ldb or (k if trans_b else n)
    ldc = ldc or n
    rows_b = n if trans_b else k
    rows_a = k if trans_a else m
    a_host = oracle.fill_uniform(batch * rows_a * lda, seed_t, -1.0, 1.0).reshape(batch, rows_a, lda)
    b_host = oracle.fill_uniform((1 if bcast_b else batch) * rows_b * ldb, seed_t + 1, -1.0, 1.0).reshape(-1, rows_b, ldb)
    ta, a_val = _to_dev(client, oracle, a_host, dtype)
    tb, b_val = _to_dev(client, oracle, b_host, dtype)
    a_t = TensorHandle.new(ta.handle, (batch, m, k), (k * lda, 1, lda) if trans_a else (m * lda, lda, 1), dtype)
    if trans_b:   # logical [k, n] stored [n][k]
        b_t = TensorHandle.new(tb.handle, (batch, k, n), (0 if bcast_b else n * ldb, 1, ldb), dtype)
    else:
        b_t = TensorHandle.new(tb.handle, (batch, k, n), (0 if bcast_b else k * ldb, ldb, 1), dtype)
    c_h = client.empty(batch * m * ldc * out_dtype.size())
    client._s.check(client.lib.mi355_memset(client.ctx, None, c_h.device_ptr(), 0xEE, c_h.size))
    c_t = TensorHandle.new(c_h, (batch, m, n), (m * ldc, ldc, 1), out_dtype)
    ops.matmul(client, a_t, b_t, c_t, algo=algo)
    raw = client.read_one(c_h)
    np_dt = np.float32 if out_dtype == ElemType.F32 else np.uint16
    got_all = raw.view(np_dt).reshape(batch, m, ldc)
    fp8_mfma = dtype in (ElemType.F8E4M3, ElemType.F8E5M2) and algo != N.GEMM_ALGO_GENERIC
    for b in range(batch):
        A = (a_val[b][:, :m].T if trans_a else a_val[b][:, :k]).astype(np.float64)
        Bm = b_val[0 if bcast_b else b]
        Bm = (Bm[:, :k].T if trans_b else Bm[:, :n]).astype(np.float64)
        ref = A @ Bm
        bound = np.abs(A) @ np.abs(Bm)
        tol = REL * bound
        if fp8_mfma:
            tol = tol + fp8_mfma_truncation_bound(A, Bm)
        got = _decode(oracle, got_all[b][:, :n], out_dtype)
        if out_dtype == ElemType.F32:
            err = np.abs(got - ref)
            assert np.all(err <= tol + 1e-30), (float(err.max()), float((err / (bound + 1e-30)).max()))
        else:
            ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(ref), 1e-30))) - (7 if out_dtype == ElemType.BF16 else 10))
            if out_dtype == ElemType.F16:       # below 2^-14 f16 is subnormal: the spacing stays 2^-24 (soak of late round 6: 640 x 5 x 1 f16, products of 2e-5 rounded
                ulp = np.maximum(ulp, 2.0 ** -24)   # correctly, 2.7e-8 off, failed the normal-range formula's 1.5e-8)
            bad = np.argwhere(np.abs(got - ref) > ulp + tol)
            assert len(bad) == 0, (len(bad), [(int(i), int(j), float(got[i, j]), float(ref[i, j])) for i, j in bad[:6]])
        if ldc > n:   # padding columns untouched
            pad = got_all[b][:, n:]
            assert np.all(pad.view(np.uint8) == 0xEE)


# ---- the reference's own known-answer vectors through the device path ------------------------------------
@pytest.mark.parametrize("algo", ["auto", "generic"])
def test_cmma_simple_1_golden(client, oracle, algo):
    lhs = oracle.to_f16(np.arange(256, dtype=np.float32))
    rhs = oracle.to_f16((np.arange(256) % 8).astype(np.float32))
    a = TensorHandle.new(client.create_from_slice(lhs), (16, 16), (16, 1), ElemType.F16)
    b = TensorHandle.new(client.create_from_slice(rhs), (16, 16), (1, 16), ElemType.F16)   # ColMajor B (cmma.rs:23)
    c = TensorHandle.new_contiguous((16, 16), client.empty(1024), ElemType.F32)
    ops.matmul(client, a, b, c, algo=ALGOS[algo])
    assert c.to_numpy(client).reshape(-1).tolist() == GOLD["cmma_simple_1_f16_16x16x16_nt"]


def test_cmma_tf32_golden_runs_as_exact_f32(client):
    lhs = np.arange(128, dtype=np.float32)
    rhs = (np.arange(128) % 8).astype(np.float32)
    a = TensorHandle.new(client.create_from_slice(lhs), (16, 8), (8, 1), ElemType.F32)
    b = TensorHandle.new(client.create_from_slice(rhs), (8, 16), (16, 1), ElemType.F32)
    c = TensorHandle.new_contiguous((16, 16), client.empty(1024), ElemType.F32)
    ops.matmul(client, a, b, c)
    assert c.to_numpy(client).reshape(-1).tolist() == GOLD["cmma_simple_tf32_16x16x8_nn"]


def test_cmma_strided_golden(client, oracle):
    m, n, k, tk = 16, 16, 32, 16
    i = np.arange(m * k)
    lhs = np.where((i % k) < tk, i - (i // k) * tk, 0).astype(np.float32)
    rhs = (np.arange(n * k) % 8).astype(np.float32)
    a = TensorHandle.new(client.create_from_slice(oracle.to_f16(lhs)), (16, 16), (32, 1), ElemType.F16)
    b = TensorHandle.new(client.create_from_slice(oracle.to_f16(rhs)), (16, 16), (1, 16), ElemType.F16)
    c = TensorHandle.new_contiguous((16, 16), client.empty(1024), ElemType.F32)
    ops.matmul(client, a, b, c)
    assert c.to_numpy(client).reshape(-1).tolist() == GOLD["cmma_strided_f16_16x16x16_nt"]


@pytest.mark.parametrize("dtype", [ElemType.F16, ElemType.BF16])
@pytest.mark.parametrize("m,n,k", [(128, 128, 64), (256, 128, 128), (64, 64, 64)])
def test_cmma_cube_expectation_on_mfma(client, oracle, dtype, m, n, k):
    # test_simple_cube_expected (cmma.rs:695-722) at MFMA-kernel sizes; values stay exactly
    # representable (lhs = i % 64, rhs = i % 8) so the comparison is exact like assert_eq!
    lhs = (np.arange(m * k) % 64).astype(np.float32)
    rhs = (np.arange(n * k) % 8).astype(np.float32)
    conv = oracle.to_f16 if dtype == ElemType.F16 else oracle.to_bf16
    a = TensorHandle.new(client.create_from_slice(conv(lhs)), (m, k), (k, 1), dtype)
    b = TensorHandle.new(client.create_from_slice(conv(rhs)), (k, n), (1, k), dtype)
    c = TensorHandle.new_contiguous((m, n), client.empty(m * n * 4), ElemType.F32)
    ops.matmul(client, a, b, c)
    expected = oracle.gemm(conv(lhs), conv(rhs), m, n, k, dtype_ab=int(dtype), trans_b=True)
    assert np.array_equal(c.to_numpy(client).reshape(-1), expected)


@pytest.mark.parametrize("dtype", [ElemType.F32, ElemType.BF16])
def test_cmma_manual_row_major_values(client, oracle, dtype):
    # test_cmma_manual (cmma.rs:1127-1177): (2i + j) x (3i + j), row-major B
    m, n, k = 32, 32, 16
    lhs = np.array([[2 * i + j for j in range(k)] for i in range(m)], dtype=np.float32)
    rhs = np.array([[3 * i + j for j in range(n)] for i in range(k)], dtype=np.float32)
    ta, _ = _to_dev(client, oracle, lhs, dtype)
    tb, _ = _to_dev(client, oracle, rhs, dtype)
    c = TensorHandle.new_contiguous((m, n), client.empty(m * n * 4), ElemType.F32)
    ops.matmul(client, TensorHandle.new(ta.handle, (m, k), (k, 1), dtype), TensorHandle.new(tb.handle, (k, n), (n, 1), dtype), c)
    expected = (lhs.astype(np.int64) @ rhs.astype(np.int64)).astype(np.float64)
    got = c.to_numpy(client).astype(np.float64)
    assert np.all(np.abs(got - expected) <= 0.03 * expected + 1e-9)      # the reference's 3 % (:1183)
    if dtype == ElemType.F32:
        assert np.array_equal(got, expected)


# ---- transpose-detecting structural checks (guide: "A=I with ASYMMETRIC B") --------------------------------
@pytest.mark.parametrize("algo,dtype,trans_b", [("f32", ElemType.F32, True), ("f32", ElemType.F32, False),
                                                 ("lp128", ElemType.BF16, True), ("lp128", ElemType.F16, True),
                                                 ("generic", ElemType.BF16, False)])
def test_identity_times_asymmetric(client, oracle, algo, dtype, trans_b):
    m = n = k = 256
    eye = np.eye(m, dtype=np.float32)
    bmat = (np.arange(k)[:, None] * 3 + np.arange(n)[None, :] * 7) % 251       # [k][n], asymmetric, exact in bf16
    bmat = bmat.astype(np.float32)
    ta, _ = _to_dev(client, oracle, eye, dtype)
    stored = np.ascontiguousarray(bmat.T) if trans_b else bmat
    tb, _ = _to_dev(client, oracle, stored, dtype)
    b_t = TensorHandle.new(tb.handle, (k, n), (1, k) if trans_b else (n, 1), dtype)
    c = TensorHandle.new_contiguous((m, n), client.empty(m * n * 4), ElemType.F32)
    ops.matmul(client, TensorHandle.new(ta.handle, (m, k), (k, 1), dtype), b_t, c, algo=ALGOS[algo])
    assert np.array_equal(c.to_numpy(client), bmat)


# ---- random-data parity per kernel -----------------------------------------------------------------------------
F32_CASES = [(128, 128, 32), (256, 384, 128), (512, 512, 512), (200, 136, 64), (1, 128, 64), (129, 1, 96), (64, 520, 32)]


@pytest.mark.parametrize("m,n,k", F32_CASES)
@pytest.mark.parametrize("trans_b", [True, False])
def test_f32_mfma_parity(client, oracle, m, n, k, trans_b):
    if not trans_b and n % 4:
        pytest.skip("row-major B needs N % 4 == 0 on the MFMA kernel (falls back to generic)")
    run_case(client, oracle, m, n, k, ElemType.F32, ElemType.F32, trans_b, ALGOS["f32"])


LP_CASES = [(128, 128, 64), (256, 256, 128), (384, 256, 512), (200, 136, 64), (1, 128, 128), (130, 2, 64), (64, 520, 192)]


@pytest.mark.parametrize("m,n,k", LP_CASES)
@pytest.mark.parametrize("dtype", [ElemType.BF16, ElemType.F16])
@pytest.mark.parametrize("out", ["f32", "same"])
def test_lp128_parity(client, oracle, m, n, k, dtype, out):
    run_case(client, oracle, m, n, k, dtype, ElemType.F32 if out == "f32" else dtype, True, ALGOS["lp128"])


@pytest.mark.parametrize("dtype,out,trans_b", [(ElemType.F32, ElemType.F32, False), (ElemType.F32, ElemType.F32, True),
                                               (ElemType.BF16, ElemType.F32, False), (ElemType.BF16, ElemType.BF16, True),
                                               (ElemType.F16, ElemType.F16, False)])
@pytest.mark.parametrize("m,n,k", [(65, 67, 19), (3, 5, 1), (130, 64, 100)])
def test_generic_parity_any_shape(client, oracle, dtype, out, trans_b, m, n, k):
    run_case(client, oracle, m, n, k, dtype, out, trans_b, ALGOS["generic"])


SPLITK_CASES = [(64, 1024, 4096), (1, 512, 2048), (128, 128, 8192), (200, 130, 1024), (64, 256, 576)]


@pytest.mark.parametrize("m,n,k", SPLITK_CASES)
@pytest.mark.parametrize("out", ["f32", "same"])
def test_split_k_skinny_shapes(client, oracle, m, n, k, out):
    # few 128x128 tiles + long K: the lp128 launcher cuts K into slices and folds f32 partial slabs in slice order
    run_case(client, oracle, m, n, k, ElemType.BF16, ElemType.F32 if out == "f32" else ElemType.BF16, True, ALGOS["lp128"])


@pytest.mark.parametrize("m,n,k", [(128, 14336, 1024), (96, 6144, 2048), (128, 4096, 2048), (256, 4096, 4096), (128, 3072, 1024)])
def test_lp128_on_both_sides_of_the_split_rule(client, oracle, m, n, k):
    # the launcher splits K up to 128 tiles (beyond: only when there are at least as many K-tiles as tiles and 48 of them) and
    # caps the slice count by the slab-traffic bound: 112 x 16 and 24 x 16 stay whole (the bound leaves one slice), 48 x 32,
    # 32 x 32 and 64 x 64 are split (round 3; rounds 1-2 kept the first two whole).  Same answers either way.
    run_case(client, oracle, m, n, k, ElemType.BF16, ElemType.BF16, True, ALGOS["lp128"])


def test_split_k_batched_padded_and_repeatable(client, oracle):
    run_case(client, oracle, 64, 256, 2048, ElemType.F16, ElemType.F16, True, ALGOS["lp128"], batch=2, lda=2056, ldb=2048, ldc=264)
    m, n, k = 64, 1024, 4096
    a = TensorHandle.uniform(client, (m, k), ElemType.BF16, 0x5EEDC0BE, 81, -1.0, 1.0)
    b = TensorHandle.uniform(client, (n, k), ElemType.BF16, 0x5EEDC0BE, 82, -1.0, 1.0)
    bt = TensorHandle.new(b.handle, (k, n), (1, k), ElemType.BF16)
    c = TensorHandle.new_contiguous((m, n), client.empty(m * n * 4), ElemType.F32)
    ops.matmul(client, a, bt, c, algo=ALGOS["lp128"])
    first = c.to_numpy(client).copy()
    for _ in range(10):      # slabs are folded in slice order: same bits every launch
        ops.matmul(client, a, bt, c, algo=ALGOS["lp128"])
        assert np.array_equal(c.to_numpy(client), first)


@pytest.mark.parametrize("m,n,k,batch,trans_b", [(1024, 1024, 1024, 1, True), (1024, 1024, 1024, 1, False), (512, 512, 4096, 1, True), (256, 384, 2048, 2, True),
                                                 (200, 1000, 1536, 1, False), (128, 128, 8192, 1, True), (1024, 512, 544, 1, True)])
def test_f32_split_k_when_the_tiles_cannot_fill_the_chip(client, oracle, m, n, k, batch, trans_b):
    """Round 4: the 128 x 128 f32 kernel cuts K when its tiles leave at least half the CUs idle (1024^3: 64 tiles, 74.5 -> ~25 us) --
    f32 slabs folded in slice order by the same fold kernel as the 16-bit path.  Parity against the f64 oracle through AUTO and
    the forced kernel; padded C; the same bits from launch to launch."""
    ldc = n + 4 if (m, n) == (200, 1000) else None
    run_case(client, oracle, m, n, k, ElemType.F32, ElemType.F32, trans_b, ALGOS["f32"], batch=batch, ldc=ldc)
    run_case(client, oracle, m, n, k, ElemType.F32, ElemType.F32, trans_b, ALGOS["auto"], batch=batch)
    a = TensorHandle.uniform(client, (batch, m, k), ElemType.F32, 0x5EEDC0BE, 85, -1.0, 1.0)
    b = TensorHandle.uniform(client, (batch, n, k) if trans_b else (batch, k, n), ElemType.F32, 0x5EEDC0BE, 86, -1.0, 1.0)
    bt = TensorHandle.new(b.handle, (batch, k, n), (n * k, 1, k) if trans_b else (k * n, n, 1), ElemType.F32)
    c = TensorHandle.new_contiguous((batch, m, n), client.empty(batch * m * n * 4), ElemType.F32)
    ops.matmul(client, a, bt, c, algo=ALGOS["f32"])
    first = c.to_numpy(client).copy()
    for _ in range(4):
        ops.matmul(client, a, bt, c, algo=ALGOS["f32"])
        assert np.array_equal(c.to_numpy(client), first)


def test_split_k_bits_do_not_depend_on_where_c_sits(client, oracle):
    """advisor, round 3: the fold of the split-K slabs had two summation trees and picked one by the ALIGNMENT of C (and by
    ldc % 4), so the same product written into a pitched or offset C could differ in the last bit.  The form now follows
    (m, n, splits) only: a C that starts 4 bytes off a 16-byte boundary with an odd pitch receives the bits of the aligned launch."""
    import ctypes as C
    m, n, k = 64, 1024, 4096
    a = TensorHandle.uniform(client, (m, k), ElemType.BF16, 0x5EEDC0BE, 83, -1.0, 1.0)
    b = TensorHandle.uniform(client, (n, k), ElemType.BF16, 0x5EEDC0BE, 84, -1.0, 1.0)
    for dt_c, np_t, esz in ((N.DTYPE_F32, np.uint32, 4), (N.DTYPE_BF16, np.uint16, 2)):
        d = N.GemmDesc(m=m, n=n, k=k, batch=1, lda=k, ldb=k, ldc=n, stride_a=m * k, stride_b=n * k, stride_c=m * n, dtype_ab=N.DTYPE_BF16,
                       dtype_c=dt_c, trans_a=0, trans_b=1, algo=N.GEMM_ALGO_LP_128)
        plan = ops.gemm_split_plan(client, d) if hasattr(ops, "gemm_split_plan") else None
        c0 = client.empty(m * n * esz)
        client._s.check(client.lib.mi355_gemm(client.ctx, None, C.byref(d), C.c_void_p(a.device_ptr()), C.c_void_p(b.device_ptr()), C.c_void_p(c0.device_ptr())))
        ref = client.read_one(c0).view(np_t).reshape(m, n)
        ldc = n + 3                                                      # odd pitch: rows start on every residue of 16 bytes
        c1 = client.empty((m * ldc + 8) * esz)
        client._s.check(client.lib.mi355_memset(client.ctx, None, C.c_void_p(c1.device_ptr()), 0xEE, c1.size))
        d.ldc, d.stride_c = ldc, m * ldc
        client._s.check(client.lib.mi355_gemm(client.ctx, None, C.byref(d), C.c_void_p(a.device_ptr()), C.c_void_p(b.device_ptr()),
                                              C.c_void_p(c1.device_ptr() + esz)))       # ... and C itself one element off the allocation
        raw = client.read_one(c1).view(np_t)
        got = raw[1:1 + m * ldc].reshape(m, ldc)
        assert np.array_equal(got[:, :n], ref)
        fill = 0xEEEEEEEE if esz == 4 else 0xEEEE
        assert np.all(got[:-1, n:] == fill) and raw[0] == fill          # the padding and the element in front of C are untouched
        assert plan is None or plan > 1


W4_CASES = [(256, 256, 64), (256, 256, 128), (256, 512, 192), (512, 512, 512), (768, 256, 1024), (512, 1024, 320),
            (256, 256, 2048)]


@pytest.mark.parametrize("m,n,k", W4_CASES)
@pytest.mark.parametrize("dtype", [ElemType.BF16, ElemType.F16])
@pytest.mark.parametrize("out", ["f32", "same"])
def test_lp256w4_parity(client, oracle, m, n, k, dtype, out):
    run_case(client, oracle, m, n, k, dtype, ElemType.F32 if out == "f32" else dtype, True, ALGOS["lp256w4"])


@pytest.mark.parametrize("m,n,k", [(256, 256, 32), (256, 256, 64), (256, 512, 96), (512, 512, 512), (768, 256, 1024),
                                   (512, 1024, 160)])
@pytest.mark.parametrize("trans_b", [True, False])
def test_lp256w4_f32_parity(client, oracle, m, n, k, trans_b):
    # f32 inputs on v_mfma_f32_32x32x2_f32: exact-f32 products, an fmaf chain in a permuted k order;
    # trans_b False = row-major B [K][N] (the DMA'd K-tile is then 32 k-rows x 256 n)
    run_case(client, oracle, m, n, k, ElemType.F32, ElemType.F32, trans_b, ALGOS["lp256w4"])


def test_lp256w4_f32_identity_batch_padding(client, oracle):
    run_case(client, oracle, 256, 256, 64, ElemType.F32, ElemType.F32, True, ALGOS["lp256w4"], batch=3)
    run_case(client, oracle, 512, 256, 32, ElemType.F32, ElemType.F32, True, ALGOS["lp256w4"], batch=2, bcast_b=True,
             lda=36, ldb=32, ldc=260)
    m = n = k = 512
    eye = np.eye(m, dtype=np.float32)
    bmat = ((np.arange(k)[:, None] * 3 + np.arange(n)[None, :] * 7) % 1021).astype(np.float32) + 0.25   # asymmetric
    ta = TensorHandle.from_numpy(client, eye)
    tb = TensorHandle.from_numpy(client, np.ascontiguousarray(bmat.T))
    c = TensorHandle.new_contiguous((m, n), client.empty(m * n * 4), ElemType.F32)
    ops.matmul(client, TensorHandle.new(ta.handle, (m, k), (k, 1), ElemType.F32),
               TensorHandle.new(tb.handle, (k, n), (1, k), ElemType.F32), c, algo=ALGOS["lp256w4"])
    assert np.array_equal(c.to_numpy(client), bmat)      # exact: one non-zero product per output
    tbn = TensorHandle.from_numpy(client, bmat)           # the same product with row-major B
    ops.matmul(client, TensorHandle.new(ta.handle, (m, k), (k, 1), ElemType.F32),
               TensorHandle.new(tbn.handle, (k, n), (n, 1), ElemType.F32), c, algo=ALGOS["lp256w4"])
    assert np.array_equal(c.to_numpy(client), bmat)
    run_case(client, oracle, 256, 512, 64, ElemType.F32, ElemType.F32, False, ALGOS["lp256w4"], batch=2, lda=68, ldb=516, ldc=512)


def test_lp256w4_identity_batch_and_fallback(client, oracle):
    run_case(client, oracle, 256, 256, 128, ElemType.BF16, ElemType.BF16, True, ALGOS["lp256w4"], batch=3)
    run_case(client, oracle, 512, 256, 64, ElemType.BF16, ElemType.F32, True, ALGOS["lp256w4"], batch=2, bcast_b=True,
             lda=72, ldb=64, ldc=260)
    m = n = k = 512
    eye = np.eye(m, dtype=np.float32)
    bmat = ((np.arange(k)[:, None] * 3 + np.arange(n)[None, :] * 7) % 251).astype(np.float32)   # asymmetric
    ta, _ = _to_dev(client, oracle, eye, ElemType.BF16)
    tb, _ = _to_dev(client, oracle, np.ascontiguousarray(bmat.T), ElemType.BF16)
    c = TensorHandle.new_contiguous((m, n), client.empty(m * n * 4), ElemType.F32)
    ops.matmul(client, TensorHandle.new(ta.handle, (m, k), (k, 1), ElemType.BF16),
               TensorHandle.new(tb.handle, (k, n), (1, k), ElemType.BF16), c, algo=ALGOS["lp256w4"])
    assert np.array_equal(c.to_numpy(client), bmat)
    # K must be a multiple of the K-tile
    with pytest.raises(ServerError) as e:
        run_case(client, oracle, 256, 256, 96, ElemType.BF16, ElemType.F32, True, ALGOS["lp256w4"])
    assert e.value.code == N.E_UNSUPPORTED


W4_RAGGED = [(300, 260, 128), (1, 256, 64), (257, 255, 320), (255, 513, 192), (700, 40, 256), (8, 8, 64), (513, 1000, 128)]


@pytest.mark.parametrize("m,n,k", W4_RAGGED)
@pytest.mark.parametrize("dtype,out", [(ElemType.BF16, "f32"), (ElemType.BF16, "same"), (ElemType.F16, "same"), (ElemType.F32, "f32")])
def test_lp256w4_ragged_edges(client, oracle, m, n, k, dtype, out):
    # edge tiles clamp their loads and skip stores outside the matrix; padded ldc so that untouched padding is checked
    if out == "same" and n % 8:       # 16-bit C rows must stay 16-byte aligned: pad the leading dimension
        ldc = (n + 7) // 8 * 8 + 8
    elif out == "f32" and n % 4:
        ldc = (n + 3) // 4 * 4 + 4
    else:
        ldc = n + (8 if out == "same" else 4)
    run_case(client, oracle, m, n, k if dtype != ElemType.F32 else k // 2, dtype, ElemType.F32 if out == "f32" else dtype, True,
             ALGOS["lp256w4"], ldc=ldc)


@pytest.mark.parametrize("m,n,k", [(300, 261, 128), (257, 255, 320), (513, 1001, 128), (256, 256, 64), (130, 770, 192)])
@pytest.mark.parametrize("dtype,out", [(ElemType.BF16, "f32"), (ElemType.BF16, "same"), (ElemType.F16, "same"), (ElemType.F32, "f32")])
@pytest.mark.parametrize("pad", [0, 1, 3])
def test_lp256w4_c_rows_off_the_16_byte_grid(client, oracle, m, n, k, dtype, out, pad):
    """C rows that start anywhere (N = 50257 logits, a view at an odd column): the epilogue stores element-wise; values against
    the oracle, the 0xEE padding of a pitched C untouched."""
    ldc = n + pad
    if (ldc * (4 if out == "f32" else 2)) % 16 == 0:
        ldc += 1
    run_case(client, oracle, m, n, k if dtype != ElemType.F32 else k // 2, dtype, ElemType.F32 if out == "f32" else dtype, True,
             ALGOS["lp256w4"], ldc=ldc, batch=2 if m < 300 else 1)


# ---- the 256 x 192 tile of the 4-wave kernel (gemm_lp256w4.hip NJ = 3; round 5) ------------------------------------------------
X192_CASES = [(256, 192, 64), (192, 192, 64), (384, 384, 128), (256, 384, 128), (512, 576, 512), (300, 200, 128), (1, 192, 64), (257, 191, 320), (255, 385, 192), (193, 100, 128),
              (700, 40, 256), (8, 8, 64), (513, 1000, 128), (768, 960, 1024), (3072, 3072, 256)]


@pytest.mark.parametrize("tile", ["lp256x192", "lp192x192"])
@pytest.mark.parametrize("m,n,k", X192_CASES)
@pytest.mark.parametrize("dtype,out", [(ElemType.BF16, "f32"), (ElemType.BF16, "same"), (ElemType.F16, "same"), (ElemType.F16, "f32")])
def test_lp256x192_parity_and_the_bits_of_the_square_tile(client, oracle, m, n, k, dtype, out, tile):
    """The 256 x 192 tile against the oracle on whole and ragged grids (pitched C: the 0xEE padding stays), and against the
    256 x 256 tile of the same kernel bit for bit -- every output element sums the same K-tiles through the same MFMAs in the
    same order, only the tile it lives in differs."""
    out_dtype = ElemType.F32 if out == "f32" else dtype
    ldc = (n + 7) // 8 * 8 + 8
    run_case(client, oracle, m, n, k, dtype, out_dtype, True, ALGOS[tile], ldc=ldc)
    a = TensorHandle.uniform(client, (m, k), dtype, 5, 1, -1.0, 1.0)
    b = TensorHandle.uniform(client, (n, k), dtype, 5, 2, -1.0, 1.0)
    bt = TensorHandle.new(b.handle, (k, n), (1, k), dtype)
    got = []
    for algo in (tile, "lp256w4"):
        c = TensorHandle.new_contiguous((m, n), client.empty(m * n * out_dtype.size()), out_dtype)
        ops.matmul(client, a, bt, c, algo=ALGOS[algo])
        got.append(client.read_one(c.handle).copy())
    assert np.array_equal(got[0], got[1])


@pytest.mark.parametrize("tile", ["lp256x192", "lp192x192"])
@pytest.mark.parametrize("pad", [1, 3])
def test_lp256x192_c_rows_off_the_16_byte_grid_batches_and_refusals(client, oracle, pad, tile):
    run_case(client, oracle, 300, 261, 128, ElemType.BF16, ElemType.BF16, True, ALGOS[tile], ldc=261 + pad, batch=2)
    run_case(client, oracle, 513, 1001, 128, ElemType.BF16, ElemType.F32, True, ALGOS[tile], ldc=1001 + pad)
    run_case(client, oracle, 256, 384, 192, ElemType.F16, ElemType.F16, True, ALGOS[tile], batch=3, bcast_b=True, lda=200, ldb=208)
    for kw in ({"dtype": ElemType.F32}, {"trans_b": False, "n": 196}, {"k": 96}):   # f32 operands, row-major B whose rows are off the 16-byte grid, K off the K-tile grid
        with pytest.raises(ServerError):
            run_case(client, oracle, 256, kw.get("n", 192), kw.get("k", 128), kw.get("dtype", ElemType.BF16), ElemType.F32, kw.get("trans_b", True), ALGOS[tile])


X192_NN_CASES = [(256, 192, 64), (192, 192, 128), (384, 384, 128), (512, 576, 512), (300, 200, 128), (1, 192, 64), (257, 184, 320), (255, 392, 192), (193, 104, 128),
                 (700, 40, 256), (8, 8, 64), (513, 1000, 128), (768, 960, 1024), (3072, 3072, 256)]


@pytest.mark.parametrize("tile", ["lp256x192", "lp192x192"])
@pytest.mark.parametrize("m,n,k", X192_NN_CASES)
@pytest.mark.parametrize("dtype,out", [(ElemType.BF16, "f32"), (ElemType.F16, "same")])
def test_lp256x192_row_major_rhs_parity_and_the_bits_of_the_square_tile(client, oracle, m, n, k, dtype, out, tile):
    """Row-major [K][N] rhs (the reference's default layout) through the narrow tiles: the transposing-read image has six
    32-column blocks per block row instead of eight.  Against the oracle (pitched C and B), and bit for bit against the
    256 x 256 tile of the same kernel on the same row-major operand."""
    out_dtype = ElemType.F32 if out == "f32" else dtype
    ldc = (n + 7) // 8 * 8 + 8
    run_case(client, oracle, m, n, k, dtype, out_dtype, False, ALGOS[tile], ldc=ldc, ldb=n + 8)
    a = TensorHandle.uniform(client, (m, k), dtype, 5, 1, -1.0, 1.0)
    b = TensorHandle.uniform(client, (k, n), dtype, 5, 2, -1.0, 1.0)
    got = []
    for algo in (tile, "lp256w4"):
        c = TensorHandle.new_contiguous((m, n), client.empty(m * n * out_dtype.size()), out_dtype)
        ops.matmul(client, a, b, c, algo=ALGOS[algo])
        got.append(client.read_one(c.handle).copy())
    assert np.array_equal(got[0], got[1])


# ---- the 256 x 256 tile on v_mfma_f32_16x16x32 (gemm_lp256m16.hip; round 5) -----------------------------------------------------
M16_CASES = [(256, 256, 64), (256, 256, 128), (256, 512, 192), (512, 512, 512), (768, 256, 1024), (512, 1024, 320), (256, 256, 2048),
             (300, 260, 128), (1, 256, 64), (257, 255, 320), (255, 513, 192), (700, 40, 256), (8, 8, 64), (513, 1000, 128)]


@pytest.mark.parametrize("m,n,k", M16_CASES)
@pytest.mark.parametrize("dtype,out", [(ElemType.BF16, "f32"), (ElemType.BF16, "same"), (ElemType.F16, "same"), (ElemType.F16, "f32")])
def test_lp256m16_parity(client, oracle, m, n, k, dtype, out):
    """The 16x16x32 form of the 256 x 256 kernel against the oracle: whole and ragged grids, pitched C (the 0xEE padding stays)."""
    ldc = (n + 7) // 8 * 8 + 8
    run_case(client, oracle, m, n, k, dtype, ElemType.F32 if out == "f32" else dtype, True, ALGOS["lp256m16"], ldc=ldc)


@pytest.mark.parametrize("pad", [1, 3])
def test_lp256m16_c_rows_off_the_16_byte_grid_batches_identity_and_refusals(client, oracle, pad):
    run_case(client, oracle, 300, 261, 128, ElemType.BF16, ElemType.BF16, True, ALGOS["lp256m16"], ldc=261 + pad, batch=2)
    run_case(client, oracle, 513, 1001, 128, ElemType.BF16, ElemType.F32, True, ALGOS["lp256m16"], ldc=1001 + pad)
    run_case(client, oracle, 256, 512, 192, ElemType.F16, ElemType.F16, True, ALGOS["lp256m16"], batch=3, bcast_b=True, lda=200, ldb=208)
    m = n = k = 512                                                      # identity x B returns B's bits: every fragment lands where it belongs
    eye = np.eye(m, dtype=np.float32)
    bmat = ((np.arange(k)[:, None] * 3 + np.arange(n)[None, :] * 7) % 251).astype(np.float32)
    ta, _ = _to_dev(client, oracle, eye, ElemType.BF16)
    tb, _ = _to_dev(client, oracle, np.ascontiguousarray(bmat.T), ElemType.BF16)
    c = TensorHandle.new_contiguous((m, n), client.empty(m * n * 4), ElemType.F32)
    ops.matmul(client, TensorHandle.new(ta.handle, (m, k), (k, 1), ElemType.BF16), TensorHandle.new(tb.handle, (k, n), (1, k), ElemType.BF16), c,
               algo=ALGOS["lp256m16"])
    assert np.array_equal(c.to_numpy(client), bmat)
    for kw in ({"dtype": ElemType.F32}, {"trans_b": False}, {"k": 96}):
        with pytest.raises(ServerError):
            run_case(client, oracle, 256, 256, kw.get("k", 128), kw.get("dtype", ElemType.BF16), ElemType.F32, kw.get("trans_b", True), ALGOS["lp256m16"])


def test_unaligned_c_gives_the_bits_of_the_aligned_form_and_stays_inside_its_rows(client, oracle):
    """4100 x 4100 x 512 bf16 through AUTO (289 tiles -> the 256x256 kernel): C placed one element into an allocation with a
    row pitch of 4101 gives the bits of the aligned product, and neither the element before it, the pitch column nor the tail of
    the allocation is written."""
    import ctypes as C
    m = n = 4100; k = 512
    a_host = oracle.fill_uniform(m * k, 91, -1.0, 1.0).reshape(m, k)
    b_host = oracle.fill_uniform(n * k, 92, -1.0, 1.0).reshape(n, k)
    ta, _ = _to_dev(client, oracle, a_host, ElemType.BF16)
    tb, _ = _to_dev(client, oracle, b_host, ElemType.BF16)
    ldc0 = 4104
    c0 = client.empty(m * ldc0 * 2)
    ldc1 = 4101
    c1 = client.empty((1 + m * ldc1 + 7) * 2)
    client._s.check(client.lib.mi355_memset(client.ctx, None, c1.device_ptr(), 0xEE, c1.size))
    # (forced: since the cost tables cover several rounds down to K = 512, AUTO hands this descriptor to the 256 x 192 tile -- the narrow tiles'
    #  own unaligned-C cases are test_lp256x192_c_rows_off_the_16_byte_grid_batches_and_refusals)
    d = N.GemmDesc(m=m, n=n, k=k, batch=1, lda=k, ldb=k, ldc=ldc0, dtype_ab=N.DTYPE_BF16, dtype_c=N.DTYPE_BF16, trans_b=1, algo=N.GEMM_ALGO_LP_256W4)
    client._s.check(client.lib.mi355_gemm(client.ctx, None, C.byref(d), C.c_void_p(ta.handle.device_ptr()),
                                          C.c_void_p(tb.handle.device_ptr()), C.c_void_p(c0.device_ptr())))
    d.ldc = ldc1
    client._s.check(client.lib.mi355_gemm(client.ctx, None, C.byref(d), C.c_void_p(ta.handle.device_ptr()),
                                          C.c_void_p(tb.handle.device_ptr()), C.c_void_p(c1.device_ptr() + 2)))
    want = client.read_one(c0).view(np.uint16).reshape(m, ldc0)[:, :n]
    raw = client.read_one(c1).view(np.uint16)
    assert raw[0] == 0xEEEE and np.all(raw[1 + m * ldc1 - 1:] == 0xEEEE)
    got = raw[1:1 + m * ldc1].reshape(m, ldc1)
    assert np.array_equal(got[:, :n], want)
    assert np.all(got[:-1, n:] == 0xEEEE)


def test_lp256w4_ragged_row_major_b_and_batch(client, oracle):
    run_case(client, oracle, 300, 260, 64, ElemType.F32, ElemType.F32, False, ALGOS["lp256w4"])
    run_case(client, oracle, 513, 1000, 96, ElemType.F32, ElemType.F32, False, ALGOS["lp256w4"], batch=2, ldb=1004, ldc=1000)
    run_case(client, oracle, 257, 255, 128, ElemType.BF16, ElemType.BF16, True, ALGOS["lp256w4"], batch=3, ldc=256)


# ---- the persistent form: one workgroup per CU walks several tiles with a continuous K-tile stream ----------------
@pytest.mark.parametrize("m,n,k", [(256, 256, 128), (256, 512, 192), (512, 512, 512), (768, 256, 1024), (512, 1024, 320)])
@pytest.mark.parametrize("dtype,out", [(ElemType.BF16, "f32"), (ElemType.BF16, "same"), (ElemType.F16, "same"), (ElemType.F32, "f32")])
def test_lp256p_parity(client, oracle, m, n, k, dtype, out):
    run_case(client, oracle, m, n, k if dtype != ElemType.F32 else k // 2, dtype, ElemType.F32 if out == "f32" else dtype, True,
             ALGOS["lp256p"])


@pytest.mark.parametrize("m,n,k,batch,dtype,out", [
    (4352, 4352, 128, 1, ElemType.BF16, "same"),     # 289 tiles on 256 CUs: 33 workgroups take a second tile, K = 2 K-tiles
    (4352, 4352, 192, 1, ElemType.BF16, "f32"),
    (256, 256, 256, 300, ElemType.BF16, "same"),     # tiles of different batch entries in one workgroup's sequence
    (512, 256, 128, 260, ElemType.F16, "f32"),
    (4352, 4352, 64, 1, ElemType.F32, "f32"),
    (256, 768, 2048, 130, ElemType.BF16, "same"),    # 390 tiles, long K
])
def test_lp256p_multi_tile_sequences(client, oracle, m, n, k, batch, dtype, out):
    run_case(client, oracle, m, n, k, dtype, ElemType.F32 if out == "f32" else dtype, True, ALGOS["lp256p"], batch=batch)


def test_lp256p_row_major_b_f32_and_padding_and_identity(client, oracle):
    run_case(client, oracle, 4352, 4352, 64, ElemType.F32, ElemType.F32, False, ALGOS["lp256p"])
    run_case(client, oracle, 512, 256, 128, ElemType.BF16, ElemType.BF16, True, ALGOS["lp256p"], batch=2, bcast_b=True,
             lda=136, ldb=128, ldc=264)
    m = n = 4352
    k = 4352
    eye = np.zeros((m, k), dtype=np.uint16)
    eye[np.arange(m), np.arange(m)] = 0x3F80
    bmat = oracle.to_bf16(oracle.fill_uniform(n * k, 91, -1.0, 1.0)).reshape(n, k)
    ta = TensorHandle.from_numpy(client, eye, ElemType.BF16)
    tb = TensorHandle.from_numpy(client, bmat, ElemType.BF16)
    c = TensorHandle.new_contiguous((m, n), client.empty(m * n * 2), ElemType.BF16)
    ops.matmul(client, TensorHandle.new(ta.handle, (m, k), (k, 1), ElemType.BF16),
               TensorHandle.new(tb.handle, (k, n), (1, k), ElemType.BF16), c, algo=ALGOS["lp256p"])
    assert np.array_equal(c.to_numpy(client).reshape(m, n), bmat.T)       # every tile of every round, bit for bit
    with pytest.raises(ServerError) as e:                                   # K must hold two K-tiles
        run_case(client, oracle, 256, 256, 64, ElemType.BF16, ElemType.F32, True, ALGOS["lp256p"])
    assert e.value.code == N.E_UNSUPPORTED


def test_lp256p_race_screen_across_tile_boundaries(client, oracle):
    # 289 tiles on 256 CUs, K = 8 K-tiles: the hand-over from one tile's stream to the next must give the same bits
    # on every launch, and the same bits as the one-tile-per-workgroup kernel (identical per-tile arithmetic order)
    m = n = 4352
    k = 512
    a = TensorHandle.uniform(client, (m, k), ElemType.BF16, 0x5EEDC0BE, 73, -1.0, 1.0)
    b = TensorHandle.uniform(client, (n, k), ElemType.BF16, 0x5EEDC0BE, 74, -1.0, 1.0)
    bt = TensorHandle.new(b.handle, (k, n), (1, k), ElemType.BF16)
    ref = TensorHandle.new_contiguous((m, n), client.empty(m * n * 4), ElemType.F32)
    ops.matmul(client, a, bt, ref, algo=ALGOS["lp256w4"])
    want = ref.to_numpy(client).copy()
    c = TensorHandle.new_contiguous((m, n), client.empty(m * n * 4), ElemType.F32)
    for _ in range(15):
        client._s.check(client.lib.mi355_memset(client.ctx, None, c.device_ptr(), 0xEE, m * n * 4))
        ops.matmul(client, a, bt, c, algo=ALGOS["lp256p"])
        assert np.array_equal(c.to_numpy(client), want)


# ---- the persistent form with dripped stores (16-bit C held in registers) -----------------------------------------------
# Same tiles, same per-tile summation order as the one-tile-per-workgroup kernel: every comparison is bit for bit.  The K values
# walk the four drip rates (8 / 4 / 2 / 1 stores per K-tile: 6-8 / 9-14 / 15-26 / 27+ K-tiles) and their edges; the shapes put
# 1, 2 and "some 1, some 2" tiles on a workgroup (the held tile of a workgroup's LAST tile leaves through the flush path).
@pytest.mark.parametrize("k", [384, 448, 512, 576, 640, 896, 960, 1024, 1664, 1728, 2048, 4160])
@pytest.mark.parametrize("m,n,batch", [(512, 512, 1), (4352, 4352, 1), (1024, 512, 72)])
def test_lp256q_is_bit_identical_to_the_plain_kernel(client, oracle, m, n, batch, k):
    if (m, k) == (4352, 4160):
        pytest.skip("covered by the smaller shapes")
    a = TensorHandle.uniform(client, (batch, m, k), ElemType.BF16, 0x5EEDC0BE, 81, -1.0, 1.0)
    b = TensorHandle.uniform(client, (batch, n, k), ElemType.BF16, 0x5EEDC0BE, 82, -1.0, 1.0)
    bt = TensorHandle.new(b.handle, (batch, k, n), (n * k, 1, k), ElemType.BF16)
    ref = TensorHandle.new_contiguous((batch, m, n), client.empty(batch * m * n * 2), ElemType.BF16)
    ops.matmul(client, a, bt, ref, algo=ALGOS["lp256w4"])
    want = ref.to_numpy(client).copy()
    c = TensorHandle.new_contiguous((batch, m, n), client.empty(batch * m * n * 2), ElemType.BF16)
    for _ in range(4):                                                     # the counted waits must hold on every launch
        client._s.check(client.lib.mi355_memset(client.ctx, None, c.device_ptr(), 0xEE, batch * m * n * 2))
        ops.matmul(client, a, bt, c, algo=ALGOS["lp256q"])
        assert np.array_equal(c.to_numpy(client), want)


def test_lp256q_oracle_f16_identity_pitched_c_and_refusals(client, oracle):
    run_case(client, oracle, 512, 768, 512, ElemType.F16, ElemType.F16, True, ALGOS["lp256q"])            # against the oracle itself
    run_case(client, oracle, 768, 512, 1024, ElemType.BF16, ElemType.BF16, True, ALGOS["lp256q"], batch=3, ldc=520)   # pitched C rows
    m = n = k = 4352                                                        # I x B^T: every tile of both rounds returns the operand's bits
    eye = np.zeros((m, k), dtype=np.uint16)
    eye[np.arange(m), np.arange(m)] = 0x3F80
    bmat = oracle.to_bf16(oracle.fill_uniform(n * k, 92, -1.0, 1.0)).reshape(n, k)
    ta, tb = TensorHandle.from_numpy(client, eye, ElemType.BF16), TensorHandle.from_numpy(client, bmat, ElemType.BF16)
    c = TensorHandle.new_contiguous((m, n), client.empty(m * n * 2), ElemType.BF16)
    ops.matmul(client, TensorHandle.new(ta.handle, (m, k), (k, 1), ElemType.BF16), TensorHandle.new(tb.handle, (k, n), (1, k), ElemType.BF16),
               c, algo=ALGOS["lp256q"])
    assert np.array_equal(c.to_numpy(client).reshape(m, n), bmat.T)
    for kw in (dict(k=320), dict(out=ElemType.F32), dict(lda=1032)):        # fewer than 6 K-tiles / f32 C / lda != ldb: refused, not mis-run
        with pytest.raises(ServerError) as e:
            run_case(client, oracle, 512, 512, kw.get("k", 1024), ElemType.BF16, kw.get("out", ElemType.BF16), True, ALGOS["lp256q"],
                     **({"lda": kw["lda"]} if "lda" in kw else {}))
        assert e.value.code == N.E_UNSUPPORTED


# ---- the persistent dripped-store form on v_mfma_f32_16x16x32 (round 6, gemm_lp256qm.hip: config C5) ---------------------------------
# Every output element is the same chain of 16x16x32 MFMAs as in gemm_lp256m16.hip (the B tile's rows are permuted on their way into
# LDS, the products are not): bit for bit against that kernel.  K walks the four drip rates and their edges as above; the shapes put
# 1, 2 and "some 1, some 2" tiles on a workgroup, so the first tile (stores branched over), the steady state and the flush all run.
@pytest.mark.parametrize("k", [384, 448, 512, 576, 640, 896, 960, 1024, 1664, 1728, 2048, 4160])
@pytest.mark.parametrize("m,n,batch", [(512, 512, 1), (4352, 4352, 1), (1024, 512, 72)])
def test_lp256qm_is_bit_identical_to_the_m16_kernel(client, oracle, m, n, batch, k):
    if (m, k) == (4352, 4160):
        pytest.skip("covered by the smaller shapes")
    a = TensorHandle.uniform(client, (batch, m, k), ElemType.BF16, 0x5EEDC0BE, 83, -1.0, 1.0)
    b = TensorHandle.uniform(client, (batch, n, k), ElemType.BF16, 0x5EEDC0BE, 84, -1.0, 1.0)
    bt = TensorHandle.new(b.handle, (batch, k, n), (n * k, 1, k), ElemType.BF16)
    ref = TensorHandle.new_contiguous((batch, m, n), client.empty(batch * m * n * 2), ElemType.BF16)
    ops.matmul(client, a, bt, ref, algo=ALGOS["lp256m16"])
    want = ref.to_numpy(client).copy()
    c = TensorHandle.new_contiguous((batch, m, n), client.empty(batch * m * n * 2), ElemType.BF16)
    for _ in range(4):                                                     # the counted waits must hold on every launch
        client._s.check(client.lib.mi355_memset(client.ctx, None, c.device_ptr(), 0xEE, batch * m * n * 2))
        ops.matmul(client, a, bt, c, algo=ALGOS["lp256qm"])
        assert np.array_equal(c.to_numpy(client), want)


@pytest.mark.parametrize("dtype", [ElemType.BF16, ElemType.F16])
@pytest.mark.parametrize("m,n,k,batch", [(512, 768, 448, 3), (768, 768, 1024, 2), (512, 512, 2048, 6), (4352, 4352, 640, 1), (1024, 512, 1728, 40)])
def test_lp256qm_row_major_b_gives_the_bits_of_the_k_contiguous_form(client, oracle, dtype, m, n, k, batch):
    """Round 6: the row-major [K][N] rhs (TensorHandle::new_contiguous, crates/cubecl-std/src/tensor/handle.rs:89) on the persistent 16x16x32
    kernel -- transposing reads of a half-swapped block image, natural column order, v_permlane16_swap in the held tile -- runs the same MFMA
    chains as the [N][K] form on the transposed matrix: bit for bit, at every drip rate, B one matrix broadcast over the batch; and one slab
    of rows against the oracle directly so that the pair cannot be wrong together."""
    a_host = oracle.fill_uniform(batch * m * k, 63, -1.0, 1.0)
    b_host = oracle.fill_uniform(k * n, 64, -1.0, 1.0).reshape(k, n)
    conv = oracle.to_bf16 if dtype == ElemType.BF16 else oracle.to_f16
    a = TensorHandle.from_numpy(client, conv(a_host), dtype)
    b_kn = TensorHandle.from_numpy(client, conv(b_host), dtype)
    b_nk = TensorHandle.from_numpy(client, conv(np.ascontiguousarray(b_host.T)), dtype)
    a_t = TensorHandle.new(a.handle, (batch, m, k), (m * k, k, 1), dtype)
    outs = []
    for handle, strides in ((b_kn, (0, n, 1)), (b_nk, (0, 1, k))):
        c = TensorHandle.new_contiguous((batch, m, n), client.empty(batch * m * n * dtype.size()), dtype)
        client._s.check(client.lib.mi355_memset(client.ctx, None, c.device_ptr(), 0xEE, batch * m * n * dtype.size()))
        ops.matmul(client, a_t, TensorHandle.new(handle.handle, (batch, k, n), strides, dtype), c, algo=ALGOS["lp256qm"])
        outs.append(c.to_numpy(client))
    assert np.array_equal(outs[0], outs[1])
    dec = oracle.from_bf16 if dtype == ElemType.BF16 else oracle.from_f16
    A = dec(conv(a_host[: m * k])).reshape(m, k)[:80].astype(np.float64)
    Bv = dec(conv(b_host)).reshape(k, n).astype(np.float64)
    got = _decode(oracle, outs[0].reshape(batch, m, n)[0][:80], dtype)
    tol = REL * (np.abs(A) @ np.abs(Bv)) + np.abs(A @ Bv) * 2.0 ** (-7 if dtype == ElemType.BF16 else -10)
    assert np.all(np.abs(got - A @ Bv) <= tol + 1e-30)


def test_lp256qm_oracle_f16_identity_pitched_operands_and_refusals(client, oracle):
    run_case(client, oracle, 512, 768, 512, ElemType.F16, ElemType.F16, True, ALGOS["lp256qm"])           # against the oracle itself
    run_case(client, oracle, 768, 512, 1024, ElemType.BF16, ElemType.BF16, True, ALGOS["lp256qm"], batch=3, ldc=520)   # pitched C rows
    run_case(client, oracle, 512, 768, 2048, ElemType.BF16, ElemType.BF16, True, ALGOS["lp256qm"], batch=2, lda=2056, ldb=2112)   # lda != ldb (one drip per K-tile)
    run_case(client, oracle, 512, 512, 640, ElemType.BF16, ElemType.BF16, True, ALGOS["lp256qm"], batch=5, bcast_b=True)          # B broadcast over the batch
    m = n = k = 4352                                                        # I x B^T: every tile of both rounds returns the operand's bits
    eye = np.zeros((m, k), dtype=np.uint16)                                 # (a wrong row of the permuted B tile shows as a wrong COLUMN here)
    eye[np.arange(m), np.arange(m)] = 0x3F80
    bmat = oracle.to_bf16(oracle.fill_uniform(n * k, 93, -1.0, 1.0)).reshape(n, k)
    ta, tb = TensorHandle.from_numpy(client, eye, ElemType.BF16), TensorHandle.from_numpy(client, bmat, ElemType.BF16)
    c = TensorHandle.new_contiguous((m, n), client.empty(m * n * 2), ElemType.BF16)
    ops.matmul(client, TensorHandle.new(ta.handle, (m, k), (k, 1), ElemType.BF16), TensorHandle.new(tb.handle, (k, n), (1, k), ElemType.BF16),
               c, algo=ALGOS["lp256qm"])
    assert np.array_equal(c.to_numpy(client).reshape(m, n), bmat.T)
    # A x I: the permutation must also hold from the other side (row i of C = row i of A)
    amat = oracle.to_bf16(oracle.fill_uniform(m * k, 94, -1.0, 1.0)).reshape(m, k)
    ta, tb = TensorHandle.from_numpy(client, amat, ElemType.BF16), TensorHandle.from_numpy(client, eye, ElemType.BF16)
    ops.matmul(client, TensorHandle.new(ta.handle, (m, k), (k, 1), ElemType.BF16), TensorHandle.new(tb.handle, (k, n), (1, k), ElemType.BF16),
               c, algo=ALGOS["lp256qm"])
    assert np.array_equal(c.to_numpy(client).reshape(m, n), amat)
    run_case(client, oracle, 512, 768, 2048, ElemType.BF16, ElemType.BF16, False, ALGOS["lp256qm"], batch=2, lda=2056, ldb=800, ldc=776)   # row-major B, pitched everything
    for kw in (dict(k=320), dict(out=ElemType.F32), dict(m=520)):   # < 6 K-tiles / f32 C / ragged tiles: refused, not mis-run
        with pytest.raises(ServerError) as e:
            run_case(client, oracle, kw.get("m", 512), 512, kw.get("k", 1024), ElemType.BF16, kw.get("out", ElemType.BF16), kw.get("trans_b", True), ALGOS["lp256qm"])
        assert e.value.code == N.E_UNSUPPORTED


@pytest.mark.parametrize("algo", ["lp128", "lp256w4", "lp256p", "f32"])     # (lp256q: its own bit-identity test above)
def test_race_screen_bitwise_repeatability(client, oracle, algo):
    # the counted-vmcnt / barrier pipeline must give the same bits on every launch (guide: "place reads by
    # the vmcnt/barrier count, never by clean runs") -- 25 launches at a multi-wave-per-CU size
    m = n = 2048
    k = 1024
    dtype = ElemType.F32 if algo == "f32" else ElemType.BF16
    a = TensorHandle.uniform(client, (m, k), dtype, 0x5EEDC0BE, 71, -1.0, 1.0)
    b = TensorHandle.uniform(client, (n, k), dtype, 0x5EEDC0BE, 72, -1.0, 1.0)
    bt = TensorHandle.new(b.handle, (k, n), (1, k), dtype)
    c = TensorHandle.new_contiguous((m, n), client.empty(m * n * 4), ElemType.F32)
    ops.matmul(client, a, bt, c, algo=ALGOS[algo])
    first = c.to_numpy(client).copy()
    ref = TensorHandle.new_contiguous((m, n), client.empty(m * n * 4), ElemType.F32)
    ops.matmul(client, a, bt, ref, algo=ALGOS["generic"])
    bound = 1e-5 * k  # |a||b| <= 1 per product
    assert np.max(np.abs(first - ref.to_numpy(client))) <= bound
    for _ in range(25):
        ops.matmul(client, a, bt, c, algo=ALGOS[algo])
        assert np.array_equal(c.to_numpy(client), first)


# ---- the 256 x 128 tile (gemm_lp128.hip with four row blocks per wave, three-stage ring + loader waves; round 3) ----------
@pytest.mark.parametrize("m,n,k,batch", [(256, 128, 64, 1),        # one tile, one K-tile
                                          (512, 256, 128, 1),       # two K-tiles: the prologue covers the whole K
                                          (2048, 2048, 448, 1),     # 128 tiles, seven K-tiles: the ring wraps twice
                                          (300, 200, 192, 3),       # ragged M and N, batch
                                          (1000, 1160, 320, 1)])    # ragged, every wave position at an edge
@pytest.mark.parametrize("dtype,out,trans_b", [(ElemType.BF16, "f32", True), (ElemType.BF16, "same", True), (ElemType.F16, "same", True),
                                                (ElemType.BF16, "f32", False), (ElemType.F16, "same", False)])
def test_lp256x128_matches_the_oracle(client, oracle, m, n, k, batch, dtype, out, trans_b):
    odt = ElemType.F32 if out == "f32" else dtype
    ldb = None if trans_b else (n + 7) // 8 * 8
    run_case(client, oracle, m, n, k, dtype, odt, trans_b, ALGOS["lp256x128"], batch=batch, ldb=ldb)


def test_lp256x128_gives_the_bits_of_the_128_tile_kernel(client, oracle):
    """Same MFMA, same k order per output, same K-tile order: the two tile heights of gemm_lp128.hip must agree bit for bit."""
    m, n, k = 1536, 1280, 1024
    a = TensorHandle.uniform(client, (m, k), ElemType.BF16, 0x5EEDC0BE, 81, -1.0, 1.0)
    b = TensorHandle.uniform(client, (n, k), ElemType.BF16, 0x5EEDC0BE, 82, -1.0, 1.0)
    bt = TensorHandle.new(b.handle, (k, n), (1, k), ElemType.BF16)
    outs = []
    for algo in ("lp128", "lp256x128"):
        c = TensorHandle.new_contiguous((m, n), client.empty(m * n * 4), ElemType.F32)
        ops.matmul(client, a, bt, c, algo=ALGOS[algo])
        outs.append(c.to_numpy(client))
    assert np.array_equal(outs[0], outs[1])
    for _ in range(10):                                   # the counted-vmcnt ring gives the same bits on every launch
        c = TensorHandle.new_contiguous((m, n), client.empty(m * n * 4), ElemType.F32)
        ops.matmul(client, a, bt, c, algo=ALGOS["lp256x128"])
        assert np.array_equal(c.to_numpy(client), outs[1])


# ---- row-major B (the reference's default rhs layout), staged natively by the tile kernels (round 3) --------------------
def _nn_desc(m, n, k, dtype, out, ldb=None, batch=1, algo=N.GEMM_ALGO_AUTO):
    ldb = ldb or n
    return N.GemmDesc(m=m, n=n, k=k, batch=batch, lda=k, ldb=ldb, ldc=n, stride_a=m * k, stride_b=k * ldb, stride_c=m * n,
                      dtype_ab=int(dtype), dtype_c=int(out), trans_a=0, trans_b=0, algo=algo)


@pytest.mark.parametrize("m,n,k,ldb", [(3072, 3072, 128, None),      # 144 tiles: one K-tile pair, every wave position
                                       (3000, 3080, 192, 3088),      # ragged M and N (N % 8 == 0), pitched rows of B
                                       (2304, 4104, 64, None),       # a single K-tile; the last tile column is 8 wide
                                       (4096, 2304, 320, 2304)])     # five K-tiles: the ring wraps
@pytest.mark.parametrize("dtype,out", [(ElemType.BF16, "f32"), (ElemType.BF16, "same"), (ElemType.F16, "f32"), (ElemType.F16, "same")])
def test_row_major_b_16bit_is_staged_natively_by_the_256_tile_kernel(client, oracle, m, n, k, ldb, dtype, out):
    """TensorHandle::new_contiguous lays a rhs out [K][N] (crates/cubecl-std/src/tensor/handle.rs:89; the reference's row-major
    cmma case is runtime_tests/cmma.rs:1160-1177).  From one round of 256x256 tiles up no operand is copied: the descriptor
    resolves to the 256x256 kernel and the re-layout plan is empty; values against the f64 oracle like every other layout."""
    odt = ElemType.F32 if out == "f32" else dtype
    d = _nn_desc(m, n, k, dtype, odt, ldb)
    # (one or two K-tiles: since late round 5 the 128 x 128 kernel's single-stage form takes them in this layout too -- 3392 x 2752 x 128 7.9 us
    #  against 10.5 -- still without a copy; the 256 x 256 kernel's form is then covered forced)
    # (with an f32 C the square tile starts at 177 tiles outside the cost tables since late round 6 -- SQUARE_MIN_TILES_F32_C: these 144 / 156 tiles take the 128 x 128 kernel)
    small = k <= 128 or out == "f32"
    assert ops.gemm_select(client, d) == (N.GEMM_ALGO_LP_128 if small else N.GEMM_ALGO_LP_256W4)
    assert ops.gemm_relayout_plan(client, d) == (False, False)
    run_case(client, oracle, m, n, k, dtype, odt, False, ALGOS["auto"], ldb=ldb)
    if small:
        run_case(client, oracle, m, n, k, dtype, odt, False, ALGOS["lp256w4"], ldb=ldb)


@pytest.mark.parametrize("dtype", [ElemType.BF16, ElemType.F16])
@pytest.mark.parametrize("algo,m,n,k,batch,out16", [
    ("lp256w4", 520, 776, 448, 3, False),      # edge tiles, broadcast B
    ("lp256w4", 512, 768, 64, 2, True),        # a single K-tile, 16-bit C
    ("lp256p", 512, 768, 448, 3, False),       # persistent form: the K-tile stream runs across tiles
    ("lp256p", 768, 512, 128, 5, True),
    ("lp256q", 512, 768, 448, 3, True),        # dripped stores, 8 per K-tile
    ("lp256q", 768, 768, 1024, 2, True),       # ... 2 per K-tile
    ("lp256q", 512, 512, 2048, 6, True),       # ... 1 per K-tile (the C5 form)
    ("lp128", 520, 776, 448, 1, False),        # 4-stage ring + loader waves (one workgroup per CU at most), edge tiles
    ("lp128", 2048, 2304, 320, 2, True),       # two-stage form + loader waves
    ("lp128", 4096, 4104, 64, 1, True),        # single-stage form (four workgroups per CU)
    ("lp128", 136, 264, 8192, 1, False),       # split-K slices
])
def test_row_major_b_gives_the_bits_of_the_k_contiguous_form(client, oracle, dtype, algo, m, n, k, batch, out16):
    """Same tile, same k order inside every MFMA, same K-tile order: every row-major-B instantiation must reproduce its
    [N][K] twin bit for bit -- forced per kernel, so that each form of each kernel is covered (the [N][K] twins are held
    to the f64 oracle by the tests above); B is one matrix broadcast over the batch."""
    a_host = oracle.fill_uniform(batch * m * k, 61, -1.0, 1.0)
    b_host = oracle.fill_uniform(k * n, 62, -1.0, 1.0).reshape(k, n)
    conv = oracle.to_bf16 if dtype == ElemType.BF16 else oracle.to_f16
    a = TensorHandle.from_numpy(client, conv(a_host), dtype)
    b_kn = TensorHandle.from_numpy(client, conv(b_host), dtype)
    b_nk = TensorHandle.from_numpy(client, conv(np.ascontiguousarray(b_host.T)), dtype)
    a_t = TensorHandle.new(a.handle, (batch, m, k), (m * k, k, 1), dtype)
    odt = dtype if out16 else ElemType.F32
    outs = []
    for handle, strides in ((b_kn, (0, n, 1)), (b_nk, (0, 1, k))):
        c = TensorHandle.new_contiguous((batch, m, n), client.empty(batch * m * n * odt.size()), odt)
        ops.matmul(client, a_t, TensorHandle.new(handle.handle, (batch, k, n), strides, dtype), c, algo=ALGOS[algo])
        outs.append(c.to_numpy(client))
    assert np.array_equal(outs[0], outs[1])
    # and one matrix against the oracle directly, so that the pair cannot be wrong together
    A = (oracle.from_bf16(conv(a_host[: m * k])) if dtype == ElemType.BF16 else oracle.from_f16(conv(a_host[: m * k]))).reshape(m, k)[:64].astype(np.float64)
    Bv = (oracle.from_bf16(conv(b_host)) if dtype == ElemType.BF16 else oracle.from_f16(conv(b_host))).reshape(k, n).astype(np.float64)
    got = _decode(oracle, outs[0].reshape(batch, m, n)[0][:64], odt)
    tol = REL * (np.abs(A) @ np.abs(Bv)) + (0 if not out16 else np.abs(A @ Bv) * 2.0 ** (-7 if dtype == ElemType.BF16 else -10))
    assert np.all(np.abs(got - A @ Bv) <= tol + 1e-30)


@pytest.mark.parametrize("m,n,k", [(1024, 1024, 1024), (2048, 2048, 2048), (200, 4096, 512), (65, 8200, 256)])
def test_row_major_b_mid_size_shapes_are_native_on_the_128_tile_kernel(client, oracle, m, n, k):
    """Below one round of 256x256 tiles AUTO lands on the 128x128 kernel, which stages row-major B itself from 65 rows up."""
    d = _nn_desc(m, n, k, ElemType.BF16, ElemType.BF16)
    assert ops.gemm_select(client, d) == N.GEMM_ALGO_LP_128 and ops.gemm_relayout_plan(client, d) == (False, False)
    run_case(client, oracle, m, n, k, ElemType.BF16, ElemType.BF16, False, ALGOS["auto"])


def test_row_major_b_through_the_strip_split_of_a_partly_filled_round(client, oracle):
    """18 x 16 = 288 tiles of 256x256 (1.125 rounds) with a long K: AUTO cuts a strip off and splits its K (gemm.cpp plan_tail_split).  With row-major B
    the strip's K slices start k rows further down B (not k columns further along its rows) and a strip of columns starts at a column
    offset: same cut, same slabs, same fold -- the bits of the [N][K] launch.  (Until round 5 this ran on 4352 x 4096 x 2048; the cost
    table now hands that [N][K] descriptor to a 192 x 192 tile, launched whole -- checked below; it prices the square tile WITH the split, which
    keeps 4608 x 4096 x 8192: 240.9 us against 265.0 on 256 x 192.  Late round 6: the split's main part -- whole rounds of full tiles -- runs on the
    persistent 16x16x32 kernel, the table prices that (x 0.95), and 6144^3 comes back from the 256 x 192 tile: 326.9 us against 338.6.)"""
    along, extent, splits = C.c_int32(), C.c_int64(), C.c_int32()
    d_nt = N.GemmDesc(m=4352, n=4096, k=2048, batch=1, lda=2048, ldb=2048, ldc=4096, dtype_ab=N.DTYPE_BF16, dtype_c=N.DTYPE_BF16, trans_b=1)
    assert ops.gemm_select(client, d_nt) == N.GEMM_ALGO_LP_192X192
    assert client.lib.mi355_gemm_tail_plan(C.byref(d_nt), C.byref(along), C.byref(extent), C.byref(splits)) == N.OK and splits.value == 1
    m, n, k = 4608, 4096, 8192
    d6 = N.GemmDesc(m=6144, n=6144, k=6144, batch=1, lda=6144, ldb=6144, ldc=6144, dtype_ab=N.DTYPE_BF16, dtype_c=N.DTYPE_BF16, trans_b=1)
    assert ops.gemm_select(client, d6) == N.GEMM_ALGO_LP_256W4
    assert client.lib.mi355_gemm_tail_plan(C.byref(d6), C.byref(along), C.byref(extent), C.byref(splits)) == N.OK and splits.value > 1
    a = TensorHandle.uniform(client, (m, k), ElemType.BF16, 0x5EEDC0BE, 91, -1.0, 1.0)
    b_nk = TensorHandle.uniform(client, (n, k), ElemType.BF16, 0x5EEDC0BE, 92, -1.0, 1.0)
    b_kn = ops.into_contiguous(client, TensorHandle.new(b_nk.handle, (k, n), (1, k), ElemType.BF16))
    outs = []
    for d, b_t in ((N.GemmDesc(m=m, n=n, k=k, batch=1, lda=k, ldb=k, ldc=n, dtype_ab=N.DTYPE_BF16, dtype_c=N.DTYPE_BF16, trans_b=1),
                    TensorHandle.new(b_nk.handle, (k, n), (1, k), ElemType.BF16)), (_nn_desc(m, n, k, ElemType.BF16, ElemType.BF16), b_kn)):
        assert ops.gemm_select(client, d) == N.GEMM_ALGO_LP_256W4
        assert client.lib.mi355_gemm_tail_plan(C.byref(d), C.byref(along), C.byref(extent), C.byref(splits)) == N.OK and splits.value > 1
        c = TensorHandle.new_contiguous((m, n), client.empty(m * n * 2), ElemType.BF16)
        ops.matmul(client, a, b_t, c)
        outs.append(c.to_numpy(client))
    assert np.array_equal(outs[0], outs[1])
    # and the split launch against the plain one (f32 slabs folded in slice order: within the 16-bit rounding of the output)
    c = TensorHandle.new_contiguous((m, n), client.empty(m * n * 2), ElemType.BF16)
    ops.matmul(client, a, b_kn, c, algo=N.GEMM_ALGO_LP_256W4)
    diff = np.abs(oracle.from_bf16(outs[1]).astype(np.float64) - oracle.from_bf16(c.to_numpy(client)).astype(np.float64))
    assert np.all(diff <= 2.0 ** -7 * np.maximum(np.abs(oracle.from_bf16(outs[1])), 1e-3) + 1e-5 * k)


@pytest.mark.parametrize("dtype", [ElemType.BF16, ElemType.F16])
@pytest.mark.parametrize("m,n,k,nn", [(4608, 4096, 8192, False), (4096, 4352, 4096, False), (4608, 4096, 8192, True)])
def test_leftover_round_split_runs_its_main_part_on_the_persistent_16x16x32_kernel(client, oracle, m, n, k, nn, dtype):
    """Late round 6 (gemm.cpp run_tail_split): the main part of a split -- whole rounds of full tiles -- is launched on gemm_lp256qm.hip where that
    kernel takes it.  The rows (columns) of the main part are then, bit for bit, what a forced launch of that kernel writes for the sub-problem on
    the same operands; the strip -- K slices on gemm_lp256w4.hip, f32 slabs folded in slice order -- agrees with a plain launch within the output's rounding."""
    along, extent, splits = C.c_int32(), C.c_int64(), C.c_int32()
    d = _nn_desc(m, n, k, dtype, dtype) if nn else N.GemmDesc(m=m, n=n, k=k, batch=1, lda=k, ldb=k, ldc=n, dtype_ab=int(dtype), dtype_c=int(dtype), trans_b=1)
    assert ops.gemm_select(client, d) == N.GEMM_ALGO_LP_256W4
    assert client.lib.mi355_gemm_tail_plan(C.byref(d), C.byref(along), C.byref(extent), C.byref(splits)) == N.OK and splits.value > 1 and extent.value > 0
    a = TensorHandle.uniform(client, (m, k), dtype, 0x5EEDC0BE, 93, -1.0, 1.0)
    b = TensorHandle.uniform(client, (k, n) if nn else (n, k), dtype, 0x5EEDC0BE, 94, -1.0, 1.0)
    b_t = b if nn else TensorHandle.new(b.handle, (k, n), (1, k), dtype)
    c = TensorHandle.new_contiguous((m, n), client.empty(m * n * 2), dtype)
    ops.matmul(client, a, b_t, c)
    whole = c.to_numpy(client).view(np.uint16).reshape(m, n)
    # the sub-problem of the main part, forced on the persistent kernel, written into a fresh C of the full pitch
    mm, mn = (extent.value, n) if along.value else (m, extent.value)
    c2 = client.empty(m * n * 2)
    d2 = N.GemmDesc(m=mm, n=mn, k=k, batch=1, lda=k, ldb=(n if nn else k), ldc=n, dtype_ab=int(dtype), dtype_c=int(dtype), trans_b=0 if nn else 1,
                    algo=N.GEMM_ALGO_LP_256QM)
    client._s.check(client.lib.mi355_gemm(client.ctx, None, C.byref(d2), C.c_void_p(a.handle.device_ptr()), C.c_void_p(b.handle.device_ptr()),
                                          C.c_void_p(c2.device_ptr())))
    main = TensorHandle.new_contiguous((m, n), c2, dtype).to_numpy(client).view(np.uint16).reshape(m, n)
    assert np.array_equal(whole[:mm, :mn], main[:mm, :mn])
    # the strip against a plain launch of the square tile
    c3 = TensorHandle.new_contiguous((m, n), client.empty(m * n * 2), dtype)
    ops.matmul(client, a, b_t, c3, algo=N.GEMM_ALGO_LP_256W4)
    plain = c3.to_numpy(client).view(np.uint16).reshape(m, n)
    dec = (lambda x: oracle.from_bf16(x)) if dtype == ElemType.BF16 else (lambda x: oracle.from_f16(x))
    strip = (slice(mm, m), slice(0, n)) if along.value else (slice(0, m), slice(mn, n))
    got, ref = dec(np.ascontiguousarray(whole[strip])).astype(np.float64), dec(np.ascontiguousarray(plain[strip])).astype(np.float64)
    assert np.all(np.abs(got - ref) <= 2.0 ** (-7 if dtype == ElemType.BF16 else -10) * np.maximum(np.abs(ref), 1e-3) + 1e-5 * k)


@pytest.mark.parametrize("m,n,k,batch,trans_b", [(512, 128, 512, 32, True), (512, 128, 512, 32, False), (128, 1024, 512, 16, True), (1024, 96, 1024, 16, False),
                                                   (1000, 72, 512, 16, True)])
def test_batched_thin_products_on_one_round_of_192_tiles(client, oracle, m, n, k, batch, trans_b):
    """Late round 6 (gemm.cpp select, the "thin, 65-128" rule over batches): heads x [seq x 128 x seq] and its transposes take the 192 x 192 tile where the
    128 x 128 kernel would split K -- AUTO's choice is checked and its output held to the oracle, batch by batch."""
    d = N.GemmDesc(m=m, n=n, k=k, batch=batch, lda=k, ldb=k if trans_b else n, ldc=n, stride_a=m * k, stride_b=n * k, stride_c=m * n,
                   dtype_ab=N.DTYPE_BF16, dtype_c=N.DTYPE_BF16, trans_b=1 if trans_b else 0)
    assert ops.gemm_select(client, d) == N.GEMM_ALGO_LP_192X192
    run_case(client, oracle, m, n, k, ElemType.BF16, ElemType.BF16, trans_b, ALGOS["auto"], batch=batch)


def test_row_major_b_refusals_of_the_tile_kernel(client, oracle):
    """N not a multiple of 8 (a 16-byte DMA piece would straddle the row end) or rows of B not 16-byte aligned: the 256x256
    kernel refuses when forced, AUTO re-lays B out and still lands on an MFMA kernel."""
    for (m, n, k, ldb) in ((3072, 3076, 128, 3080), (3072, 3072, 128, 3076)):
        d = _nn_desc(m, n, k, ElemType.BF16, ElemType.F32, ldb, algo=N.GEMM_ALGO_LP_256W4)
        a = client.empty(m * k * 2)
        b = client.empty(k * ldb * 2)
        c = client.empty(m * n * 4)
        rc = client.lib.mi355_gemm(client.ctx, None, C.byref(d), C.c_void_p(a.device_ptr()), C.c_void_p(b.device_ptr()), C.c_void_p(c.device_ptr()))
        assert rc == N.E_UNSUPPORTED
        d.algo = N.GEMM_ALGO_AUTO
        assert ops.gemm_relayout_plan(client, d) == (False, True) and ops.gemm_select(client, d) != N.GEMM_ALGO_GENERIC
    # few COLUMNS: B is the small operand, re-laid out for the streaming kernel (which only exists for K-contiguous operands);
    # few ROWS (x [M][K] times a row-major weight [K][N], the decode case): B is the streamed operand and is never transposed --
    # the strip kernel (up to 8 rows at this size, round 4) or the 128x128 kernel stage it natively
    d = _nn_desc(8192, 16, 8192, ElemType.BF16, ElemType.BF16)
    assert ops.gemm_relayout_plan(client, d) == (False, True) and ops.gemm_select(client, d) == N.GEMM_ALGO_STREAM64
    for m in (1, 16, 64):
        d = _nn_desc(m, 8192, 8192, ElemType.BF16, ElemType.BF16)
        assert ops.gemm_relayout_plan(client, d) == (False, False)
        assert ops.gemm_select(client, d) == (N.GEMM_ALGO_NNROWS if m <= 16 else N.GEMM_ALGO_LP_128)


# ---- layouts the MFMA kernels do not stage directly: re-laid out K-contiguous into library scratch first ------------
@pytest.mark.parametrize("m,n,k", [(512, 512, 512), (300, 260, 128), (256, 1024, 320), (1000, 513, 192)])
@pytest.mark.parametrize("dtype,out", [(ElemType.BF16, "f32"), (ElemType.BF16, "same"), (ElemType.F16, "f32")])
def test_row_major_b_16bit_goes_through_relayout(client, oracle, m, n, k, dtype, out):
    ldb = (n + 7) // 8 * 8
    d = N.GemmDesc(m=m, n=n, k=k, batch=1, lda=k, ldb=ldb, ldc=n, dtype_ab=int(dtype), dtype_c=N.DTYPE_F32, trans_b=0)
    assert ops.gemm_select(client, d) != N.GEMM_ALGO_GENERIC            # an MFMA kernel after the transpose
    ldc = n if out == "f32" and n % 4 == 0 else (n + 7) // 8 * 8 + 8
    run_case(client, oracle, m, n, k, dtype, ElemType.F32 if out == "f32" else dtype, False, ALGOS["auto"], ldb=ldb, ldc=ldc)


def test_relayout_batched_broadcast_and_transposed_a(client, oracle):
    run_case(client, oracle, 256, 512, 256, ElemType.BF16, ElemType.F32, False, ALGOS["auto"], batch=3, ldb=512)
    run_case(client, oracle, 256, 512, 256, ElemType.BF16, ElemType.BF16, False, ALGOS["auto"], batch=3, bcast_b=True, ldb=520)
    # transposed A (stored [K][M]) x row-major B, f32 and bf16: both operands are re-laid out
    for dtype in (ElemType.F32, ElemType.BF16):
        m, n, k = 384, 512, 256
        a = oracle.fill_uniform(m * k, 61, -1.0, 1.0).reshape(m, k)
        b = oracle.fill_uniform(k * n, 62, -1.0, 1.0).reshape(k, n)
        ta, a_val = _to_dev(client, oracle, np.ascontiguousarray(a.T), dtype)        # device holds A^T: [K][M]
        tb, b_val = _to_dev(client, oracle, b, dtype)
        lhs = TensorHandle.new(ta.handle, (m, k), (1, m), dtype)                      # logical [M][K], column-major storage
        rhs = TensorHandle.new(tb.handle, (k, n), (n, 1), dtype)
        c = TensorHandle.new_contiguous((m, n), client.empty(m * n * 4), ElemType.F32)
        ops.matmul(client, lhs, rhs, c)
        A = a_val.reshape(k, m).T.astype(np.float64)
        Bm = b_val.reshape(k, n).astype(np.float64)
        ref, bound = A @ Bm, np.abs(A) @ np.abs(Bm)
        assert np.all(np.abs(c.to_numpy(client) - ref) <= REL * bound)
    # below the size threshold the generic kernel still answers (and agrees)
    run_case(client, oracle, 40, 24, 16, ElemType.BF16, ElemType.F32, False, ALGOS["auto"])


@pytest.mark.parametrize("m,n,k,dtype,trans_b", [
    (512, 512, 520, ElemType.BF16, True),      # K not a multiple of the 64-wide K-tile: zero-padded copies of A and B
    (512, 512, 1000, ElemType.F16, True),
    (300, 260, 200, ElemType.BF16, False),     # ragged K AND row-major B
    (512, 512, 100, ElemType.F32, True),       # f32: K-tile is 32
    (384, 512, 250, ElemType.F32, False),
    (512, 512, 515, ElemType.BF16, True),      # odd K: rows not even 4-byte multiples
])
def test_ragged_k_goes_through_zero_padded_relayout(client, oracle, m, n, k, dtype, trans_b):
    ldb = k if trans_b else (n + 7) // 8 * 8
    d = N.GemmDesc(m=m, n=n, k=k, batch=1, lda=k, ldb=ldb, ldc=n, dtype_ab=int(dtype), dtype_c=N.DTYPE_F32, trans_b=int(trans_b))
    assert ops.gemm_select(client, d) not in (N.GEMM_ALGO_GENERIC,)
    run_case(client, oracle, m, n, k, dtype, ElemType.F32, trans_b, ALGOS["auto"], ldb=ldb)


def test_unaligned_operand_rows_go_through_relayout(client, oracle):
    # lda / ldb that break the 16-byte row alignment the DMA needs
    run_case(client, oracle, 512, 512, 256, ElemType.BF16, ElemType.F32, True, ALGOS["auto"], lda=259, ldb=261)
    run_case(client, oracle, 512, 512, 128, ElemType.F32, ElemType.F32, True, ALGOS["auto"], lda=131, ldb=129, batch=2)


def test_padded_leading_dimensions_and_untouched_padding(client, oracle):
    run_case(client, oracle, 256, 128, 128, ElemType.F32, ElemType.F32, True, ALGOS["auto"], lda=136, ldb=132, ldc=140)
    run_case(client, oracle, 256, 128, 128, ElemType.BF16, ElemType.F32, True, ALGOS["auto"], lda=136, ldb=144, ldc=132)
    run_case(client, oracle, 128, 256, 64, ElemType.BF16, ElemType.BF16, True, ALGOS["auto"], lda=72, ldb=64, ldc=264)
    run_case(client, oracle, 128, 128, 64, ElemType.F32, ElemType.F32, False, ALGOS["auto"], lda=64, ldb=132, ldc=128)


def test_batched_and_broadcast(client, oracle):
    run_case(client, oracle, 128, 128, 64, ElemType.BF16, ElemType.F32, True, ALGOS["auto"], batch=5)
    run_case(client, oracle, 128, 256, 64, ElemType.BF16, ElemType.BF16, True, ALGOS["auto"], batch=3, bcast_b=True)
    run_case(client, oracle, 128, 128, 32, ElemType.F32, ElemType.F32, False, ALGOS["auto"], batch=4)
    run_case(client, oracle, 40, 24, 20, ElemType.F32, ElemType.F32, True, ALGOS["auto"], batch=2, bcast_b=True)


def test_auto_selection_and_errors(client):
    d = N.GemmDesc(m=4096, n=4096, k=4096, batch=1, lda=4096, ldb=4096, ldc=4096, dtype_ab=N.DTYPE_F32,
                   dtype_c=N.DTYPE_F32, trans_b=1)
    assert ops.gemm_select(client, d) == N.GEMM_ALGO_LP_256W4
    d.trans_b, d.ldb = 0, 4096
    assert ops.gemm_select(client, d) == N.GEMM_ALGO_LP_256W4
    d.m = 4096 + 64                                                    # 17 x 16 square tiles: the launcher splits the 17-tile strip's K (plan_tail_split), and the
    assert ops.gemm_select(client, d) == N.GEMM_ALGO_LP_256W4          # fitted f32 table (late round 6) prices the square tile with that split
    d.m, d.n, d.ldb, d.ldc = 4672, 3968, 3968, 3968                     # 19 x 16 ragged square tiles, no split possible: two rounds -- the 128 x 128 tile
    assert ops.gemm_select(client, d) == N.GEMM_ALGO_F32_MFMA          # (1 441 us against 1 823)
    d.m, d.n, d.ldb, d.ldc = 4096 + 64, 4096, 4096, 4096
    d.k = 4096 + 16                                                    # K not a multiple of the 32-wide f32 K-tile:
    assert ops.gemm_select(client, d) == N.GEMM_ALGO_LP_256W4          # zero-padded scratch copies, still the MFMA kernel
    d = N.GemmDesc(m=2048, n=2048, k=2048, batch=1, lda=2048, ldb=2048, ldc=2048, dtype_ab=N.DTYPE_BF16,
                   dtype_c=N.DTYPE_BF16, trans_b=1)
    assert ops.gemm_select(client, d) == N.GEMM_ALGO_LP_128             # 64 tiles of 256^2: the 128x128 kernel fills the chip better
    d.m = d.n = d.lda = d.ldb = d.ldc = d.k = 3072                      # 144 tiles of 256^2 (+55 % over the 128x128 kernel) -- and since
    assert ops.gemm_select(client, d) == N.GEMM_ALGO_LP_192X192         # round 5 256 tiles of 192^2 of the same kernel: another +25 %
    d = N.GemmDesc(m=8192, n=8192, k=8192, batch=1, lda=8192, ldb=8192, ldc=8192, dtype_ab=N.DTYPE_BF16,
                   dtype_c=N.DTYPE_BF16, trans_b=1)
    assert ops.gemm_select(client, d) == N.GEMM_ALGO_LP_256QM           # four rounds at the power limit: the 16x16x32 form (round 5), persistent with dripped stores (round 6)
    # several rounds of short tiles: the persistent form (config C5's shard: 64 x 2048^3); long K or a single round: not
    d = N.GemmDesc(m=2048, n=2048, k=2048, batch=64, lda=2048, ldb=2048, ldc=2048, stride_a=2048 * 2048, stride_b=2048 * 2048,
                   stride_c=2048 * 2048, dtype_ab=N.DTYPE_BF16, dtype_c=N.DTYPE_BF16, trans_b=1)
    assert ops.gemm_select(client, d) == N.GEMM_ALGO_LP_256QM           # 16-bit C, K = 32 K-tiles: the dripped-store form of it (on 16x16x32 MFMAs since round 6)
    d.dtype_c = N.DTYPE_F32
    assert ops.gemm_select(client, d) == N.GEMM_ALGO_LP_256M16          # f32 C cannot be held in registers: the one-tile 16x16x32 kernel (late round 6: 911 us / 927 on lp256p)
    d.dtype_c = N.DTYPE_BF16
    d.k = d.lda = d.ldb = 640
    assert ops.gemm_select(client, d) == N.GEMM_ALGO_LP_256QM           # 10 K-tiles: four stores per K-tile (round 6; the 32x32x16 form measured slower than lp256p there)
    d.k = d.lda = d.ldb = 2048
    d.batch = 4
    assert ops.gemm_select(client, d) == N.GEMM_ALGO_LP_256QM           # exactly one full round (4 x 64 tiles): the same kernel, one tile per workgroup (round 6, second K loop)
    d.batch = 3
    assert ops.gemm_select(client, d) == N.GEMM_ALGO_LP_256QM           # 192 tiles: wherever the square tile is the choice (round 6: 3584^3 80.2 -> 75.2 us)
    d = N.GemmDesc(m=8192 + 8, n=8192, k=8192, batch=1, lda=8192, ldb=8192, ldc=8192, dtype_ab=N.DTYPE_BF16,
                   dtype_c=N.DTYPE_BF16, trans_b=1)
    assert ops.gemm_select(client, d) == N.GEMM_ALGO_LP_256W4          # ragged M / N stay on the fast kernel (4.1 rounds: the launcher
    d.ldc = 8192 + 3                                                   # splits the last strip off, the 32x32x16 kernel's ground)
    assert ops.gemm_select(client, d) == N.GEMM_ALGO_LP_256W4          # C rows not 16-byte aligned: the same kernel stores element-wise
    d = N.GemmDesc(m=100, n=100, k=7, batch=1, lda=7, ldb=7, ldc=100, dtype_ab=N.DTYPE_BF16, dtype_c=N.DTYPE_F32, trans_b=1)
    assert ops.gemm_select(client, d) == N.GEMM_ALGO_GENERIC
    a = TensorHandle.new_contiguous((8, 8), client.empty(256), ElemType.F32)
    with pytest.raises(ServerError) as e:
        ops.matmul(client, a, TensorHandle.new_contiguous((4, 8), client.empty(128), ElemType.F32), a)
    assert e.value.code == N.E_INVALID_ARGUMENT
    with pytest.raises(ServerError) as e:   # lda smaller than the row
        ops.matmul(client, TensorHandle.new(a.handle, (8, 8), (4, 1), ElemType.F32), a, a)
    assert e.value.code in (N.E_UNSUPPORTED_STRIDES, N.E_INVALID_ARGUMENT)
    # K == 0 writes zeros
    c = TensorHandle.new_contiguous((8, 8), client.create_from_slice(np.ones(64, dtype=np.float32)), ElemType.F32)
    ops.matmul(client, TensorHandle.new_contiguous((8, 0), client.empty(0), ElemType.F32),
               TensorHandle.new_contiguous((0, 8), client.empty(0), ElemType.F32), c)
    assert not c.to_numpy(client).any()


# ---- OCP FP8 operands (v_mfma_f32_32x32x64_f8f6f4 inside the 256x256 kernel; generic kernel for the rest) ---------
F8 = [ElemType.F8E4M3, ElemType.F8E5M2]


@pytest.mark.parametrize("dtype", F8)
def test_fp8_fill_and_casts_match_the_oracle_bit_for_bit(client, oracle, dtype):
    n = 1 << 16
    t = TensorHandle.uniform(client, (n,), dtype, 0x5EEDC0BE, 77, -3.0, 3.0)
    want = oracle.to_fp8(oracle.fill_uniform(n, 77, -3.0, 3.0), int(dtype))
    assert np.array_equal(t.to_numpy(client), want)
    # f32 -> fp8: specials, saturation, subnormals, ties; fp8 -> f32: every encoding
    codes = np.arange(256, dtype=np.uint8)
    dec = oracle.from_fp8(codes, int(dtype))
    pos = dec[:128][np.isfinite(dec[:128])]
    mids = ((pos[:-1].astype(np.float64) + pos[1:].astype(np.float64)) / 2).astype(np.float32)
    x = np.concatenate([np.float32([0.0, -0.0, np.inf, -np.inf, np.nan, 1e30, -1e30, 448.0, 449.0, 464.0, 465.0, 57344.0, 61440.0,
                                    1e-10, -1e-10, 2.0 ** -10, 2.0 ** -17]), mids, -mids,
                        np.nextafter(mids, np.float32(0)), np.nextafter(mids, np.float32(1e9)),
                        oracle.fill_uniform(4096, 78, -500.0, 500.0)]).astype(np.float32)
    src = TensorHandle.from_numpy(client, x)
    dst = client.empty(x.size)
    client._s.check(client.lib.mi355_cast(client.ctx, None, src.device_ptr(), N.DTYPE_F32, dst.device_ptr(), int(dtype), x.size))
    got = client.read_one(dst).view(np.uint8)[: x.size]
    assert np.array_equal(got, oracle.to_fp8(x, int(dtype)))
    back = client.empty(256 * 4)
    csrc = TensorHandle.from_numpy(client, codes, dtype)
    client._s.check(client.lib.mi355_cast(client.ctx, None, csrc.device_ptr(), int(dtype), back.device_ptr(), N.DTYPE_F32, 256))
    gotf = client.read_one(back).view(np.float32)[:256]
    assert np.array_equal(np.isnan(gotf), np.isnan(dec)) and np.array_equal(gotf[~np.isnan(dec)], dec[~np.isnan(dec)])


@pytest.mark.parametrize("dtype", F8)
@pytest.mark.parametrize("m,n,k", [(3, 5, 7), (65, 67, 130), (64, 64, 256)])
def test_fp8_generic_kernel_is_bit_exact_against_the_f32_loop(client, oracle, dtype, m, n, k):
    # products of fp8 values are exact in f32, so the generic kernel's fma chain equals the reference loop
    # (test_simple_cube_expected, cmma.rs:695-722) bit for bit, in k order
    a = oracle.to_fp8(oracle.fill_uniform(m * k, 90, -1.0, 1.0), int(dtype))
    b = oracle.to_fp8(oracle.fill_uniform(n * k, 91, -1.0, 1.0), int(dtype))
    ta = TensorHandle.new(client.create_from_slice(a), (m, k), (k, 1), dtype)
    tb = TensorHandle.new(client.create_from_slice(b), (k, n), (1, k), dtype)
    c = TensorHandle.new_contiguous((m, n), client.empty(m * n * 4), ElemType.F32)
    ops.matmul(client, ta, tb, c, algo=N.GEMM_ALGO_GENERIC)
    want = oracle.gemm(a, b, m, n, k, dtype_ab=int(dtype), trans_b=True)
    assert np.array_equal(c.to_numpy(client).reshape(-1), want)


@pytest.mark.parametrize("dtype", F8)
@pytest.mark.parametrize("out", [ElemType.F32, ElemType.BF16, ElemType.F16])
@pytest.mark.parametrize("m,n,k", [(256, 256, 128), (256, 512, 384), (512, 768, 1024), (300, 504, 256), (5, 4096, 512)])
def test_fp8_mfma_parity(client, oracle, dtype, out, m, n, k):
    d = N.GemmDesc(m=m, n=n, k=k, batch=1, lda=k, ldb=k, ldc=n, dtype_ab=int(dtype), dtype_c=int(out), trans_b=1)
    assert ops.gemm_select(client, d) == N.GEMM_ALGO_LP_128                    # at most 128 tiles of 256^2: the 128x128 kernel
    run_case(client, oracle, m, n, k, dtype, out, True, N.GEMM_ALGO_AUTO)
    run_case(client, oracle, m, n, k, dtype, out, True, N.GEMM_ALGO_LP_256W4)


@pytest.mark.parametrize("dtype", F8)
@pytest.mark.parametrize("m,n,k,batch", [(128, 128, 128, 1), (128, 256, 256, 1), (384, 128, 1152, 1), (200, 333, 640, 1), (64, 8192, 4096, 1),
                                         (2048, 2048, 1024, 1), (256, 256, 512, 5), (3072, 3072, 512, 1)])
def test_fp8_128x128_kernel_every_pipeline_form(client, oracle, dtype, m, n, k, batch):
    """1, 2, 9, 5 ... K-tiles through the 2-stage and the 4-stage ring, ragged edges, the split-K slabs, batches."""
    out = ElemType.BF16 if (m + n) % 3 else ElemType.F32
    run_case(client, oracle, m, n, k, dtype, out, True, N.GEMM_ALGO_LP_128, batch=batch)
    d = N.GemmDesc(m=3072, n=3072, k=1024, batch=1, lda=1024, ldb=1024, ldc=3072, dtype_ab=int(dtype), dtype_c=N.DTYPE_BF16, trans_b=1)
    assert ops.gemm_select(client, d) == N.GEMM_ALGO_LP_256W4                  # 144 tiles of 256^2, eight K-tiles of 128 values
    d.k = d.lda = d.ldb = 512                                                  # four K-tiles: the 128x128 kernel (late round 6, the fp8 audit: level or ahead on 19 of 21
    assert ops.gemm_select(client, d) == N.GEMM_ALGO_LP_128                    # shapes past 100 square tiles at K = 512)


def test_fp8_identity_returns_operand_values_and_batches(client, oracle):
    m = n = 256
    k = 512
    for dtype, one in ((ElemType.F8E4M3, 0x38), (ElemType.F8E5M2, 0x3C)):
        eye = np.zeros((m, k), dtype=np.uint8)
        eye[np.arange(m), np.arange(m) * 2 + 1] = one                           # row i picks k = 2i + 1
        bbits = oracle.to_fp8(oracle.fill_uniform(3 * n * k, 92, -300.0, 300.0), int(dtype)).reshape(3, n, k)
        a = TensorHandle.new(client.create_from_slice(eye), (3, m, k), (0, k, 1), dtype)     # broadcast A
        b = TensorHandle.new(client.create_from_slice(bbits), (3, k, n), (n * k, 1, k), dtype)
        c = TensorHandle.new_contiguous((3, m, n), client.empty(3 * m * n * 4), ElemType.F32)
        ops.matmul(client, a, b, c, algo=N.GEMM_ALGO_LP_256W4)
        got = c.to_numpy(client)
        want = oracle.from_fp8(bbits, int(dtype))[:, :, 1::2].transpose(0, 2, 1)              # C[b][i][j] = B[b][j][2i+1]
        assert np.array_equal(got, want)


@pytest.mark.parametrize("dtype", F8)
def test_fp8_other_layouts_go_through_relayout_or_generic(client, oracle, dtype):
    # row-major B, ragged K, padded / unaligned rows, transposed A: re-laid out into scratch, then the MFMA kernel
    d = N.GemmDesc(m=512, n=512, k=512, batch=1, lda=512, ldb=512, ldc=512, dtype_ab=int(dtype), dtype_c=N.DTYPE_F32, trans_b=0)
    assert ops.gemm_select(client, d) == N.GEMM_ALGO_LP_128                   # an MFMA kernel after the transpose (4 tiles: the 128x128 one)
    run_case(client, oracle, 512, 512, 512, dtype, ElemType.F32, False, N.GEMM_ALGO_AUTO)
    run_case(client, oracle, 300, 260, 200, dtype, ElemType.F32, True, N.GEMM_ALGO_AUTO)             # ragged K
    run_case(client, oracle, 256, 256, 256, dtype, ElemType.BF16, True, N.GEMM_ALGO_AUTO, lda=259, ldb=263, ldc=272)
    run_case(client, oracle, 256, 384, 256, dtype, ElemType.F32, True, N.GEMM_ALGO_AUTO, batch=3, bcast_b=True)
    run_case(client, oracle, 16, 16, 32, dtype, ElemType.F32, True, N.GEMM_ALGO_AUTO)                # tiny: generic
    # bitwise repeatability (race screen) of the MFMA path
    a = TensorHandle.uniform(client, (1024, 2048), dtype, 1, 93, -1.0, 1.0)
    b = TensorHandle.uniform(client, (1024, 2048), dtype, 1, 94, -1.0, 1.0)
    bt = TensorHandle.new(b.handle, (2048, 1024), (1, 2048), dtype)
    c1 = TensorHandle.new_contiguous((1024, 1024), client.empty(4 << 20), ElemType.F32)
    c2 = TensorHandle.new_contiguous((1024, 1024), client.empty(4 << 20), ElemType.F32)
    ops.matmul(client, a, bt, c1)
    first = c1.to_numpy(client).copy()
    for _ in range(5):
        ops.matmul(client, a, bt, c2)
        assert np.array_equal(c2.to_numpy(client), first)


def test_fp8_rejects_what_it_does_not_do(client):
    d = dict(m=256, n=256, k=256, batch=1, lda=256, ldb=256, ldc=256, trans_b=1)
    a = client.empty(1 << 16)
    with pytest.raises(ServerError):        # fp8 output is not produced by the GEMM
        client._s.check(client.lib.mi355_gemm(client.ctx, None, C.byref(N.GemmDesc(dtype_ab=N.DTYPE_F8E4M3, dtype_c=N.DTYPE_F8E4M3, **d)),
                                              a.device_ptr(), a.device_ptr(), a.device_ptr()))


# ---- partly filled last round: main part + split-K strip (gemm.cpp plan_tail_split) ------------------------------------------
@pytest.mark.parametrize("m,n,k,dtype,out", [
    (4608, 4096, 8192, ElemType.BF16, ElemType.BF16),      # 288 tiles = 1.125 rounds, long K: strip of two tile rows, K split eight ways
    (6144, 6144, 1024, ElemType.BF16, ElemType.BF16),      # (until round 5 the split's case; now 256 x 192 tiles, launched whole)
    (5000, 3328, 2048, ElemType.BF16, ElemType.F32),       # 20 x 13 = 260 tiles, ragged M: strip of columns
    (3328, 5000, 2048, ElemType.F16, ElemType.F16),        # the transposed case, ragged N (n % 4 == 0)
    (4608, 4608, 2048, ElemType.F8E4M3, ElemType.BF16),    # fp8: 18 x 18 = 324 tiles
    (4352, 4096, 512, ElemType.F32, ElemType.F32),         # f32: 17 x 16 = 272 tiles
])
def test_partly_filled_last_round_is_split_and_still_exact_enough(client, oracle, m, n, k, dtype, out):
    run_case(client, oracle, m, n, k, dtype, out, True, N.GEMM_ALGO_AUTO)
    # the same bits from launch to launch (the fold adds the slabs in slice order)
    a = TensorHandle.uniform(client, (m, k), dtype, 1, 95, -1.0, 1.0)
    b = TensorHandle.uniform(client, (n, k), dtype, 1, 96, -1.0, 1.0)
    bt = TensorHandle.new(b.handle, (k, n), (1, k), dtype)
    c1 = TensorHandle.new_contiguous((m, n), client.empty(m * n * out.size()), out)
    c2 = TensorHandle.new_contiguous((m, n), client.empty(m * n * out.size()), out)
    ops.matmul(client, a, bt, c1)
    ops.matmul(client, a, bt, c2)
    assert np.array_equal(c1.to_numpy(client), c2.to_numpy(client))


# ---- D = A * B + C: the C operand of cmma::execute(a, b, c, d) (frontend/cmma.rs:1066-1110) --------------------
@pytest.mark.parametrize("dtype,out_dtype", [(ElemType.F32, ElemType.F32), (ElemType.BF16, ElemType.F32), (ElemType.BF16, ElemType.BF16),
                                             (ElemType.F16, ElemType.F16), (ElemType.F8E4M3, ElemType.BF16)])
@pytest.mark.parametrize("m,n,k,batch,ldc", [(16, 16, 16, 1, 16), (256, 512, 128, 1, 512), (300, 257, 96, 2, 264), (1024, 1024, 512, 3, 1024),
                                             (64, 100, 0, 1, 100)])
def test_matmul_add(client, oracle, dtype, out_dtype, m, n, k, batch, ldc):
    a_host = oracle.fill_uniform(batch * m * max(k, 1), 61, -1.0, 1.0).reshape(batch, m, max(k, 1))[:, :, :k]
    b_host = oracle.fill_uniform(batch * n * max(k, 1), 62, -1.0, 1.0).reshape(batch, n, max(k, 1))[:, :, :k]
    c_host = oracle.fill_uniform(batch * m * ldc, 63, -4.0, 4.0).reshape(batch, m, ldc)
    ta, a_val = _to_dev(client, oracle, np.ascontiguousarray(a_host), dtype)
    tb, b_val = _to_dev(client, oracle, np.ascontiguousarray(b_host), dtype)
    tc, c_val = _to_dev(client, oracle, c_host, out_dtype)
    a_t = TensorHandle.new(ta.handle, (batch, m, k), (m * k, k, 1), dtype)
    b_t = TensorHandle.new(tb.handle, (batch, k, n), (n * k, 1, k), dtype)                 # stored [n][k]
    c_t = TensorHandle.new(tc.handle, (batch, m, n), (m * ldc, ldc, 1), out_dtype)
    d_h = client.empty(batch * m * ldc * out_dtype.size())
    client._s.check(client.lib.mi355_memset(client.ctx, None, d_h.device_ptr(), 0xEE, d_h.size))
    d_t = TensorHandle.new(d_h, (batch, m, n), (m * ldc, ldc, 1), out_dtype)
    ops.matmul(client, a_t, b_t, d_t, acc=c_t)
    np_dt = np.float32 if out_dtype == ElemType.F32 else np.uint16
    got_all = client.read_one(d_h).view(np_dt).reshape(batch, m, ldc)
    for b in range(batch):
        A, Bm = a_val[b].astype(np.float64), b_val[b].astype(np.float64).T
        Cm = c_val[b][:, :n].astype(np.float64)
        ref = A @ Bm + Cm
        bound = np.abs(A) @ np.abs(Bm) + np.abs(Cm)
        got = _decode(oracle, got_all[b][:, :n], out_dtype)
        if out_dtype == ElemType.F32:
            assert np.all(np.abs(got - ref) <= REL * bound + 1e-30)
        else:
            ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(ref), 1e-30))) - (7 if out_dtype == ElemType.BF16 else 10))
            if out_dtype == ElemType.F16:       # below 2^-14 f16 is subnormal: the spacing stays 2^-24 (soak of late round 6: 640 x 5 x 1 f16, products of 2e-5 rounded
                ulp = np.maximum(ulp, 2.0 ** -24)   # correctly, 2.7e-8 off, failed the normal-range formula's 1.5e-8)
            assert np.all(np.abs(got - ref) <= ulp + REL * bound)
        if ldc > n:
            assert np.all(got_all[b][:, n:].view(np.uint8) == 0xEE)
    # in place: D aliases C
    ops.matmul(client, a_t, b_t, c_t, acc=c_t)
    assert np.array_equal(client.read_one(tc.handle).view(np_dt).reshape(batch, m, ldc)[:, :, :n], got_all[:, :, :n])


def test_matmul_add_exact_small_integers_and_oracle(client, oracle):
    # integer-valued operands: every product, the sum and the addition are exact in f32 -> bit equality with the oracle's
    # restatement (oracle.gemm_add), for f32 and for bf16 output (values below 256 are exact in bf16)
    m, n, k = 64, 48, 32
    a = (np.arange(m * k) % 5 - 2).astype(np.float32).reshape(m, k)
    b = (np.arange(n * k) % 3 - 1).astype(np.float32).reshape(n, k)
    c = (np.arange(m * n) % 7 - 3).astype(np.float32).reshape(m, n)
    for dt, odt, odt_o in ((ElemType.F32, ElemType.F32, oracle.DT_F32), (ElemType.BF16, ElemType.BF16, oracle.DT_BF16)):
        ta, _ = _to_dev(client, oracle, a, dt)
        tb, _ = _to_dev(client, oracle, b, dt)
        tc, _ = _to_dev(client, oracle, c, odt)
        b_t = TensorHandle.new(tb.handle, (k, n), (1, k), dt)
        d = TensorHandle.new_contiguous((m, n), client.empty(m * n * odt.size()), odt)
        ops.matmul(client, ta, b_t, d, acc=tc)
        a_bits = a if dt == ElemType.F32 else oracle.to_bf16(a)
        b_bits = b if dt == ElemType.F32 else oracle.to_bf16(b)
        c_bits = c if odt == ElemType.F32 else oracle.to_bf16(c)
        want = oracle.gemm_add(a_bits, b_bits, c_bits, m, n, k, dtype_ab=int(dt), dtype_c=odt_o, trans_b=True)
        assert np.array_equal(d.to_numpy(client).reshape(-1), want[: m * n])
    # acc in another layout than out (pitched) is brought into out's layout first; wrong dtype / shape are refused
    tcp = TensorHandle.new(client.create_from_slice(np.pad(c, ((0, 0), (0, 16)))), (m, n), (n + 16, 1), ElemType.F32)
    ta, _ = _to_dev(client, oracle, a, ElemType.F32)
    tb, _ = _to_dev(client, oracle, b, ElemType.F32)
    d = TensorHandle.new_contiguous((m, n), client.empty(m * n * 4), ElemType.F32)
    ops.matmul(client, ta, TensorHandle.new(tb.handle, (k, n), (1, k), ElemType.F32), d, acc=tcp)
    assert np.array_equal(d.to_numpy(client), a @ b.T + c)
    with pytest.raises(ServerError):
        ops.matmul(client, ta, TensorHandle.new(tb.handle, (k, n), (1, k), ElemType.F32), d,
                   acc=TensorHandle.new_contiguous((m, n), client.empty(m * n * 2), ElemType.BF16))


@pytest.mark.parametrize("dtype", [ElemType.BF16, ElemType.F32, ElemType.F8E4M3])
@pytest.mark.parametrize("m,n,k,batch,algo", [(256, 512, 256, 1, "lp256w4"), (300, 260, 128, 2, "lp256w4"), (300, 261, 128, 2, "lp256w4"),
                                              (4096, 2304, 256, 1, "auto"),
                                              (4000, 3500, 256, 1, "auto"), (1024, 768, 512, 5, "auto")])
def test_matmul_add_f32_inside_the_256_kernel(client, oracle, dtype, m, n, k, batch, algo):
    """f32 output on the 256x256 kernel adds C in its epilogue (no product scratch): whole and ragged tiles, batches, in place."""
    a_host = oracle.fill_uniform(batch * m * k, 71, -1.0, 1.0).reshape(batch, m, k)
    b_host = oracle.fill_uniform(batch * n * k, 72, -1.0, 1.0).reshape(batch, n, k)
    c_host = oracle.fill_uniform(batch * m * n, 73, -8.0, 8.0).reshape(batch, m, n)
    ta, a_val = _to_dev(client, oracle, a_host, dtype)
    tb, b_val = _to_dev(client, oracle, b_host, dtype)
    tc = TensorHandle.from_numpy(client, c_host)
    a_t = TensorHandle.new(ta.handle, (batch, m, k), (m * k, k, 1), dtype)
    b_t = TensorHandle.new(tb.handle, (batch, k, n), (n * k, 1, k), dtype)
    d_t = TensorHandle.new_contiguous((batch, m, n), client.empty(batch * m * n * 4), ElemType.F32)
    ops.matmul(client, a_t, b_t, d_t, algo=ALGOS[algo], acc=tc)
    got = d_t.to_numpy(client)
    rows = np.unique(np.concatenate([np.arange(0, m, 97), [m - 1, min(m - 1, 255), min(m - 1, 256)]]))
    for b in range(batch):
        A, Bm = a_val[b][rows].astype(np.float64), b_val[b].astype(np.float64).T
        ref = A @ Bm + c_host[b][rows].astype(np.float64)
        bound = np.abs(A) @ np.abs(Bm) + np.abs(c_host[b][rows]).astype(np.float64)
        assert np.all(np.abs(got[b][rows].astype(np.float64) - ref) <= REL * bound + 1e-30)
    # the plain product plus C computed on the host in f32 is the same thing up to the last f32 rounding of the addition
    p_t = TensorHandle.new_contiguous((batch, m, n), client.empty(batch * m * n * 4), ElemType.F32)
    ops.matmul(client, a_t, b_t, p_t, algo=ALGOS[algo])
    assert np.array_equal(got, p_t.to_numpy(client) + c_host)           # same kernel, same sum order: bit-equal
    ops.matmul(client, a_t, b_t, tc, algo=ALGOS[algo], acc=tc)           # in place
    assert np.array_equal(tc.to_numpy(client), got)


def test_matmul_add_f32_stays_fused_where_auto_names_the_16x16x32_kernel(client, oracle):
    """Advisor, round 5: AUTO's rule for gemm_lp256m16.hip took large [N][K] products with an f32 C off the fused epilogue -- a
    4-bytes-per-output scratch product + an add pass, and MI355_E_UNSUPPORTED inside a capture window on a stream without that scratch.
    mi355_gemm_add answers every form of the square tile with the plain kernel's fused epilogue: here inside a capture window on a fresh
    stream (nothing may be allocated), in place, against the f64 oracle on sampled rows."""
    import ctypes as C
    lib, ctx, chk = client.lib, client.ctx, client._s.check
    m = n = 8192
    k = 4160                                                   # 1 024 tiles, 65 K-tiles: past the persistent forms, AUTO names the 16x16x32 kernel
    a = TensorHandle.uniform(client, (m, k), ElemType.BF16, 1, 41, -1.0, 1.0)
    b = TensorHandle.uniform(client, (n, k), ElemType.BF16, 1, 42, -1.0, 1.0)
    cacc = TensorHandle.uniform(client, (m, n), ElemType.F32, 1, 43, -8.0, 8.0)
    c0 = cacc.to_numpy(client).reshape(m, n).copy()
    d = N.GemmDesc(m=m, n=n, k=k, batch=1, lda=k, ldb=k, ldc=n, dtype_ab=N.DTYPE_BF16, dtype_c=N.DTYPE_F32, trans_b=1)
    assert ops.gemm_select(client, d) == N.GEMM_ALGO_LP_256M16
    client.sync()
    st = C.c_void_p()
    chk(lib.mi355_stream_create(ctx, C.byref(st)))
    chk(lib.mi355_graph_begin_capture(ctx, st))
    chk(lib.mi355_gemm_add(ctx, st, C.byref(d), C.c_void_p(a.device_ptr()), C.c_void_p(b.device_ptr()), C.c_void_p(cacc.device_ptr()),
                           C.c_void_p(cacc.device_ptr())))    # in place, no scratch: must not be refused
    g = C.c_void_p()
    chk(lib.mi355_graph_end_capture(ctx, st, C.byref(g)))
    chk(lib.mi355_graph_replay(ctx, st, g))
    chk(lib.mi355_sync(ctx, st))
    got = cacc.to_numpy(client).reshape(m, n)
    rows = np.array([0, 1, 127, 128, 255, 256, 4095, 4096, 8191, 5003])
    A = oracle.from_bf16(a.to_numpy(client).reshape(m, k)[rows]).astype(np.float64)
    Bm = oracle.from_bf16(b.to_numpy(client).reshape(n, k)).astype(np.float64).T
    ref = A @ Bm + c0[rows].astype(np.float64)
    bound = np.abs(A) @ np.abs(Bm) + np.abs(c0[rows]).astype(np.float64)
    assert np.all(np.abs(got[rows].astype(np.float64) - ref) <= REL * bound + 1e-30)
    chk(lib.mi355_graph_destroy(ctx, g))
    chk(lib.mi355_stream_destroy(ctx, st))


# ---- at most 16 rows against a row-major [K][N] weight: wide row strips, register transposition, 4x4x4 MFMA (gemm_nnrows.hip) -------
# strip width by shape (plan_for): 16 x 8192 x 8192 -> 1024 B x 16 slices, x 4096 -> 512 B x 8, x 2048 -> 256 B x 4; one slice when the
# strips alone fill the chip (N = 131072); K slices of 64 ... 8192 rows = every relation of the ring depth to the iteration count
@pytest.mark.parametrize("m,n,k,kw", [
    (1, 8192, 8192, {}), (16, 8192, 8192, {}), (16, 8192, 4096, {}), (16, 8192, 2048, {}), (4, 4096, 4096, {}), (8, 2048, 8192, {}),
    (3, 8, 8, {}), (1, 8, 8200, {}), (16, 16, 64, {}), (7, 520, 1032, {}), (16, 1032, 520, {"ldc": 1040}), (2, 131072, 512, {}),
    (13, 4104, 6152, {"ldb": 4112, "lda": 6160}), (5, 8192, 1096, {"batch": 3}), (16, 3072, 16384, {}), (1, 1024, 65536, {}),
    (12, 28672, 1024, {}), (9, 2048, 2056, {"batch": 2, "ldc": 2051})])
@pytest.mark.parametrize("dtype,out_dtype", [(ElemType.BF16, ElemType.BF16), (ElemType.F16, ElemType.F32), (ElemType.BF16, ElemType.F32),
                                             (ElemType.F32, ElemType.F32)])     # f32 operands: v_mfma_f32_4x4x1, four columns per lane
def test_few_rows_times_row_major_weight_matches_the_oracle(client, oracle, m, n, k, kw, dtype, out_dtype):
    run_case(client, oracle, m, n, k, dtype, out_dtype, False, ALGOS["nnrows"], **kw)


@pytest.mark.parametrize("seed", range(int(os.environ.get("NNROWS_FUZZ_SEEDS", "40"))))
def test_few_rows_times_row_major_weight_random_shapes(client, oracle, seed):
    """Seeded draws over everything the strip kernel's geometry depends on: rows (row blocks of 4), N (strips, ragged last strip),
    K (slices, ring rounds, ragged last iteration, x chunks of 2048 / 4096 / 8192), pitches, batch, dtypes."""
    rng = np.random.default_rng(1000 + seed)
    m = int(rng.integers(1, 17))
    n = 8 * int(rng.choice([1, 3, 16, 33, 64, 129, 500, 1024, 2051, 4100]))
    k = 8 * int(rng.choice([1, 2, 7, 8, 9, 31, 64, 97, 256, 511, 1025, 2048, 3000]))
    if n * k > (1 << 26):
        k = max(8, (1 << 26) // n // 8 * 8)
    kw = {"batch": int(rng.choice([1, 1, 1, 2, 3]))}
    if rng.random() < 0.4:
        kw["ldb"] = n + 8 * int(rng.integers(1, 5))
    if rng.random() < 0.4:
        kw["lda"] = k + 8 * int(rng.integers(1, 5))
    if rng.random() < 0.4:
        kw["ldc"] = n + int(rng.integers(1, 9))
    dtype = [ElemType.BF16, ElemType.F16, ElemType.F32][int(rng.choice([0, 0, 1, 2]))]
    out = ElemType.F32 if (rng.random() < 0.5 or dtype == ElemType.F32) else dtype
    if dtype == ElemType.F32:                                  # half the element count per byte: keep the operand bytes
        k = max(8, k // 2 // 8 * 8)
        if "lda" in kw:
            kw["lda"] = k + 8
    run_case(client, oracle, m, n, k, dtype, out, False, ALGOS["nnrows"], **kw)


def test_few_rows_times_row_major_weight_selection_refusals_and_determinism(client, oracle):
    bf = N.DTYPE_BF16
    sel = lambda m, n, k, **kw: ops.gemm_select(client, _nn_desc(m, n, k, bf, bf, **kw))
    assert sel(1, 8192, 8192) == sel(8, 8192, 8192) == sel(16, 8192, 8192) == sel(4, 4096, 14336) == sel(16, 128256, 4096) == N.GEMM_ALGO_NNROWS
    # 16 rows below N = 8192; 250 column tiles need no K split; too little to stream; too many rows
    assert N.GEMM_ALGO_NNROWS not in (sel(16, 6144, 6144), sel(4, 32000, 4096), sel(4, 4096, 4096), sel(17, 131072, 4096))
    d = _nn_desc(16, 8192, 8192, bf, bf); d.trans_b = 1; d.ldb = 8192
    assert ops.gemm_select(client, d) != N.GEMM_ALGO_NNROWS                                               # [N][K] weights: the streaming kernels
    for m, n, k in [(17, 512, 512), (4, 516, 512), (4, 512, 516)]:                                         # forced: refused
        with pytest.raises(ServerError) as e:
            run_case(client, oracle, m, n, k, ElemType.BF16, ElemType.BF16, False, ALGOS["nnrows"])
        assert e.value.code == N.E_UNSUPPORTED
    # run to run bit-identical (K slices meet in slice order), and the per-strip tickets are back at zero after every call
    m, n, k = 16, 8192, 8192
    a = TensorHandle.uniform(client, (m, k), ElemType.BF16, 1, 5, -1.0, 1.0)
    b = TensorHandle.uniform(client, (k, n), ElemType.BF16, 1, 6, -1.0, 1.0)
    outs = []
    for _ in range(4):
        c = TensorHandle.new_contiguous((m, n), client.empty(m * n * 4), ElemType.F32)
        ops.matmul(client, a, b, c, algo=ALGOS["nnrows"])
        outs.append(c.to_numpy(client).copy())
    assert all(np.array_equal(outs[0], o) for o in outs[1:])
    assert np.isfinite(outs[0]).all() and np.abs(outs[0]).max() > 1.0


@pytest.mark.parametrize("elem, dtype", [(ElemType.BF16, N.DTYPE_BF16), (ElemType.F32, N.DTYPE_F32)], ids=["bf16", "f32"])
def test_few_rows_times_row_major_weight_inside_a_capture_window(client, oracle, elem, dtype):
    """The strip kernel's K-slice scratch and ticket words cannot be created inside a capture window: on a stream that has them
    (a warm-up call) the captured launch replays the eager bits; on a stream that does not, AUTO takes the tile kernel instead of
    failing (as the split-K paths fall back) -- and the replay is still right.  (f32: the advisor of round 4 found the fallback
    never left the strip kernel there -- the f32 rule did not look at the switch -- and the captured call returned E_UNSUPPORTED.)"""
    import ctypes as C
    lib, ctx, chk = client.lib, client.ctx, client._s.check
    m, n, k = 4, 8192, 8192
    a = TensorHandle.uniform(client, (m, k), elem, 1, 7, -1.0, 1.0)
    b = TensorHandle.uniform(client, (k, n), elem, 1, 8, -1.0, 1.0)
    d = _nn_desc(m, n, k, dtype, N.DTYPE_F32)
    assert ops.gemm_select(client, d) == N.GEMM_ALGO_NNROWS
    client.sync()
    want = None
    streams = [C.c_void_p(), C.c_void_p()]                    # both alive at once: a destroyed stream's handle (and scratch) may be reused
    for st in streams:
        chk(lib.mi355_stream_create(ctx, C.byref(st)))
    for warm, st in zip((True, False), streams):
        c = client.empty(m * n * 4)
        run = lambda: chk(lib.mi355_gemm(ctx, st, C.byref(d), C.c_void_p(a.device_ptr()), C.c_void_p(b.device_ptr()), C.c_void_p(c.device_ptr())))
        if warm:
            run()
            chk(lib.mi355_sync(ctx, st))
            want = client.read_one(c).view(np.float32).copy()
        chk(lib.mi355_graph_begin_capture(ctx, st))
        run()
        g = C.c_void_p()
        chk(lib.mi355_graph_end_capture(ctx, st, C.byref(g)))
        chk(lib.mi355_memset(ctx, st, C.c_void_p(c.device_ptr()), 0xEE, m * n * 4))
        chk(lib.mi355_graph_replay(ctx, st, g))
        chk(lib.mi355_sync(ctx, st))
        got = client.read_one(c).view(np.float32)
        if warm:
            assert np.array_equal(got, want)                                   # the strip kernel, same bits as eager
        else:
            assert np.allclose(got, want, rtol=0, atol=1e-3 * np.abs(want).max()) and not np.array_equal(got, want)   # another kernel's association
        chk(lib.mi355_graph_destroy(ctx, g))
    for st in streams:
        chk(lib.mi355_stream_destroy(ctx, st))


# ---- at most 16 rows or columns: the dot2 row-streaming kernel (gemm_skinny.hip) ---------------------------------------------
@pytest.mark.parametrize("m,n,k,kw", [
    (1, 8192, 8192, {}),                       # the GEMV the bench quotes
    (1, 37, 8, {}),                            # one partial chunk, ragged row count
    (2, 1000, 4104, {}),                       # K = 8 full chunks + 8 elements
    (3, 777, 520, {"lda": 528, "ldb": 536}),   # padded rows on both operands
    (4, 64, 1024, {"ldc": 72}),                # pitched C: padding stays untouched
    (5, 130, 2048, {"batch": 3}),
    (8, 256, 1544, {"batch": 2, "bcast_b": True}),
    (16, 4096, 1024, {}),                      # 64 partial sums per lane: the full butterfly
    (13, 19, 72, {}),
    (4096, 1, 1024, {}),                       # N <= 16: roles swapped, output walked column-wise
    (1000, 7, 4104, {"ldc": 8}),
    (515, 16, 640, {"batch": 2}),
])
@pytest.mark.parametrize("dtype,out_dtype", [(ElemType.BF16, ElemType.BF16), (ElemType.F16, ElemType.F32), (ElemType.BF16, ElemType.F32),
                                             (ElemType.F32, ElemType.F32)])     # f32 operands (round 4): plain FMAs, four elements per 16-byte piece
def test_skinny_dot2_kernel_matches_the_oracle(client, oracle, m, n, k, kw, dtype, out_dtype):
    d = N.GemmDesc(m=m, n=n, k=k, batch=kw.get("batch", 1), lda=kw.get("lda", k), ldb=kw.get("ldb", k), ldc=kw.get("ldc", n),
                   stride_a=m * kw.get("lda", k), stride_b=0 if kw.get("bcast_b") else n * kw.get("ldb", k), stride_c=m * kw.get("ldc", n),
                   dtype_ab=int(dtype), dtype_c=int(out_dtype), trans_a=0, trans_b=1, algo=0)
    if min(m, n) <= 2:       # AUTO: one or two rows / columns stream through this kernel, wider ones keep the MFMA path when it runs
        assert ops.gemm_select(client, d) == N.GEMM_ALGO_SKINNY
    run_case(client, oracle, m, n, k, dtype, out_dtype, True, ALGOS["skinny"], **kw)
    run_case(client, oracle, m, n, k, dtype, out_dtype, True, ALGOS["auto"], **kw)


def test_skinny_kernel_refusals_and_determinism(client, oracle):
    # both extents above 16, a row-major B, an unaligned K: refused when forced, AUTO goes elsewhere
    for (m, n, k, tb) in ((17, 17, 64, True), (4, 64, 60, True), (4, 64, 64, False)):
        with pytest.raises(ServerError):
            run_case(client, oracle, m, n, k, ElemType.BF16, ElemType.F32, tb, ALGOS["skinny"])
        run_case(client, oracle, m, n, k, ElemType.BF16, ElemType.F32, tb, ALGOS["auto"])
    # fixed summation order: repeated launches give the same bits
    a = TensorHandle.uniform(client, (8, 8192), ElemType.BF16, 3, 1, -1.0, 1.0)
    b = TensorHandle.uniform(client, (4096, 8192), ElemType.BF16, 3, 2, -1.0, 1.0)
    outs = []
    for _ in range(3):
        c = TensorHandle.new_contiguous((8, 4096), client.empty(8 * 4096 * 4), ElemType.F32)
        ops.matmul(client, a, TensorHandle.new(b.handle, (8192, 4096), (1, 8192), ElemType.BF16), c, algo=ALGOS["skinny"])
        outs.append(c.to_numpy(client).copy())
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


@pytest.mark.parametrize("k", [64, 128, 192, 256])
@pytest.mark.parametrize("m,n,kw", [(2176, 2176, {}), (2100, 2260, {"ldc": 2264}), (640, 1152, {"batch": 7})])
@pytest.mark.parametrize("dtype,out_dtype", [(ElemType.BF16, ElemType.BF16), (ElemType.F16, ElemType.F32)])
def test_lp128_single_stage_form_for_short_k(client, oracle, m, n, k, kw, dtype, out_dtype):
    """More workgroups than CUs and K of at most four K-tiles: gemm_lp128.hip runs with one LDS stage, four workgroups per CU
    (fetch / wait / multiply serialised inside a workgroup), and stores whole rows through the LDS transposition."""
    run_case(client, oracle, m, n, k, dtype, out_dtype, True, ALGOS["lp128"], **kw)


def test_output_bound_shapes_select_the_small_tile(client):
    def sel(m, n, k, batch=1):
        d = N.GemmDesc(m=m, n=n, k=k, batch=batch, lda=k, ldb=k, ldc=n, stride_a=m * k, stride_b=n * k, stride_c=m * n,
                       dtype_ab=N.DTYPE_BF16, dtype_c=N.DTYPE_BF16, trans_a=0, trans_b=1, algo=0)
        return ops.gemm_select(client, d)
    assert sel(8192, 8192, 64) == sel(8192, 8192, 256) == sel(16384, 8192, 192) == N.GEMM_ALGO_LP_128   # several rounds, K <= 192 (<= 256 up to 1280 tiles)
    assert sel(16384, 8192, 256) == sel(8192, 8192, 320) == N.GEMM_ALGO_LP_256P         # ... beyond: the persistent large tile (cold operands, round 3)
    assert sel(4096, 4096, 64) == N.GEMM_ALGO_LP_128                   # one round of 256 tiles, one K-tile: 9.9 us against 10.7 on the large tile (cold; round 5)
    assert sel(8192, 8192, 320) in (N.GEMM_ALGO_LP_256P, N.GEMM_ALGO_LP_256Q, N.GEMM_ALGO_LP_256W4)
    assert sel(1, 8192, 8192) == sel(8192, 2, 4096) == N.GEMM_ALGO_SKINNY
    assert sel(4, 8192, 8192) == sel(16, 8192, 8192) == sel(64, 8192, 8192) == sel(8192, 64, 8192) == N.GEMM_ALGO_STREAM64   # 3 ... 64 rows: no split-K
    assert sel(64, 32768, 4096) == sel(64, 4096, 16384) == sel(65, 8192, 8192) == sel(64, 64, 8192) == N.GEMM_ALGO_LP_128   # many rounds / few long workgroups / 65 rows
    assert sel(64, 1024, 2048) == sel(64, 7168, 8192) == sel(32, 8192, 16384) == sel(16, 28672, 8192) == N.GEMM_ALGO_STREAM64
    # round 4 (33-64 rows, after the 128x128 kernel's 64 x 128 tile): short grids walking a long K, more workgroups than CUs, K past 8192
    assert sel(64, 64, 4096) == sel(64, 2048, 8192) == sel(64, 14336, 4096) == sel(64, 8192, 14336) == sel(48, 4096, 4096) == N.GEMM_ALGO_LP_128
    assert sel(48, 512, 8192) == sel(64, 512, 8192) == N.GEMM_ALGO_LP_128   # rounds 3 and 4
    # (until late round 6 17-32 rows left the streaming kernel from 768 workgroups, up to 16 from 2048: profiles/r06_stream_large_grids_ab.txt -- 32 x 57344
    #  x 4096 85.8 us against 87.3, 32 x 57344 x 8192 149.9 / 157.1, 16 x 128256 x 4096 163.9 / 197.1, 32 x 28672 x 4096 level)
    assert sel(32, 57344, 4096) == sel(16, 128256, 4096) == sel(4, 152064, 8192) == N.GEMM_ALGO_STREAM64 and sel(32, 28672, 4096) == N.GEMM_ALGO_LP_128   # (224 column tiles: one round of the 128 x 128 kernel)
    assert sel(32, 90000, 4096) == sel(16, 200000, 4096) == N.GEMM_ALGO_LP_128       # past the measured grids (2560 / 4800 workgroups)
    assert sel(64, 8192, 28672) == sel(64, 28672, 8192) == N.GEMM_ALGO_LP_128        # small operand past 2 MiB / 64 rows over more than 512 workgroups
    assert sel(8192, 32, 8192) == N.GEMM_ALGO_STREAM64 and sel(8192, 32, 14336) == N.GEMM_ALGO_LP_128   # few columns: K past 8192 goes to split-K (round 4)
    assert sel(44440, 88, 1536) == sel(16384, 512, 1024) == N.GEMM_ALGO_LP_256X128 and sel(32768, 128, 1024) == N.GEMM_ALGO_LP_128   # tall and skinny (round 4)
    assert sel(4, 2048, 4096) == sel(384, 4, 8192) == sel(4096, 4, 14336) == N.GEMM_ALGO_SKINNY   # 3-4 rows, fewer than 192 streaming workgroups
    assert sel(3, 512, 14336) == sel(3, 896, 14336) == N.GEMM_ALGO_LP_128    # ... but K past 8192 on a handful of tiles: split-K (round 6 audit: 11.8 us against 17.0)
    assert sel(8192, 4, 2048) == sel(4, 8192, 8192) == N.GEMM_ALGO_STREAM64           # ... from 192 up the streaming kernel
    # the 256 x 128 tile: more than one 128x128 tile per CU, at most one 256 x 128 tile per CU, long K (round 3)
    assert sel(2048, 2048, 8192, batch=2) == N.GEMM_ALGO_LP_256X128
    # round 5: the cost table (gemm.cpp TILE_COSTS, fitted by tools/dev/tile_cost_model.py) decides among the tile kernels for [N][K]
    # operands past 512 x 512 up to one round of the square tile: one round of 192 x 192 tiles takes most of that band, 256 x 192
    # tiles where those would need a second round
    assert sel(4096, 2048, 4096) == sel(2560, 2560, 3072) == sel(2048, 3072, 8192) == sel(4096, 1536, 8192) == sel(3072, 3072, 3072) == sel(2304, 2304, 2304) == N.GEMM_ALGO_LP_192X192
    assert sel(4096, 3072, 4096) == sel(3328, 3328, 4096) == N.GEMM_ALGO_LP_256X192 and sel(3584, 3584, 3584) == N.GEMM_ALGO_LP_256QM and sel(4096, 4096, 4096) == N.GEMM_ALGO_LP_256QM   # (one FULL round: the 16x16x32 form, round 6)
    assert sel(2048, 2048, 8192) == N.GEMM_ALGO_LP_128                                  # one 128x128 tile per CU (121 tiles of 192^2: too few)
    assert sel(4096, 2048, 2048) == sel(3072, 2560, 1024) == N.GEMM_ALGO_LP_192X192       # round 5 (were the 128x128 kernel's: 1059 / 781, 918 / 693 TFLOP/s)
    assert sel(4096, 2304, 4096) == N.GEMM_ALGO_LP_256X192                            # 144 tiles of 256^2, 288 of 256 x 128 (two rounds), 192 of 256 x 192: the table's call
    assert sel(32, 512, 2048) == sel(512, 16, 2048) == sel(32, 6144, 8192) == N.GEMM_ALGO_STREAM64   # few workgroups are fine up to K = 2048; 192 at any K
    assert sel(32, 512, 8192) == sel(512, 16, 8192) == sel(16, 2048, 8192) == sel(32, 1024, 4096) == N.GEMM_ALGO_LP_128   # round 4: split-K instead
    assert sel(8192, 4096, 512) == sel(9216, 3072, 640) == N.GEMM_ALGO_LP_256QM      # 512 / 432 tiles: persistent from one round up (round 6: the 16x16x32 form at every drip rate)
    # (384 tiles at K = 1024 ... 2048: the cost tables, multi-round down to K = 512 since late round 5, take 256 x 192 tiles: profiles/
    #  r05_persistent_vs_narrow_ab.txt; at K = 512 they were within 5 % of the dripped-store form and the 16x16x32 K loop of round 6 takes it)
    assert sel(8192, 3072, 512) == N.GEMM_ALGO_LP_256QM and sel(8192, 3072, 2048) == N.GEMM_ALGO_LP_256X192
    assert sel(4096, 4096, 512) == N.GEMM_ALGO_LP_256QM                               # exactly one full round: the 16x16x32 K loop, one tile per workgroup


# ---- 3 ... 64 rows or columns: the no-split-K streaming kernel with loader waves (gemm_stream64.hip) -------------------------
@pytest.mark.parametrize("m,n,k,kw", [
    (64, 8192, 8192, {}),                        # the skinny shape the bench quotes
    (64, 1000, 2048, {}),                        # ragged streamed extent: the last workgroup has 8 of its 32 rows
    (33, 257, 1024, {"lda": 1032, "ldb": 1040}), # two row blocks with one valid row in the second; padded operand rows
    (32, 96, 64, {}),                            # a single K-tile (shorter than the ring)
    (17, 64, 704, {"ldc": 72}),                  # 11 K-tiles: one short of the 12-slot ring; pitched C
    (3, 4096, 4096, {}),
    (48, 512, 1600, {"batch": 3}),               # 25 K-tiles: ring wraps twice
    (64, 320, 1024, {"batch": 2, "bcast_b": True}),
    (8192, 64, 4096, {}),                        # N <= 64: roles swapped, output tile stored transposed
    (1000, 40, 2048, {"ldc": 48}),
    (513, 7, 640, {"batch": 2}),
    (16, 9000, 1024, {}),                        # 282 workgroups: more than one per CU -> half-depth rings, two per CU
    (64, 8200, 640, {}),                         # the same form with two row blocks; ragged last workgroup
    (24, 300, 2048, {"batch": 40}),              # 400 workgroups through the batch
    # round 5, the NB = 2 form (64 streamed rows per workgroup, K in two slices folded by the last workgroup of a row block): taken from
    # 56 rows at K >= 8192 on grids of 129-256 workgroups of the NB = 1 form -- (64, 8192, 8192) above is one
    (60, 5000, 8192, {"lda": 8200, "ldc": 5008}),   # 157 workgroups -> 79 row blocks, the last with 8 of its 64 rows; padded rows
    (4200, 57, 8192, {}),                        # roles swapped: the output block stored transposed by the folding workgroup
    (64, 4160, 16384, {}),                       # 128 K-tiles per slice
    # late round 6, the grids AUTO now streams (profiles/dispatch_rules.md STREAM_PART_TILES_*, STREAM_COLS_K_MAX): four and a half rounds of
    # workgroups with 47 rows, two and a third with 6 columns walking 224 K-tiles
    (47, 37312, 2048, {}),
    (19152, 6, 14336, {}),
    (4, 100000, 512, {}),                        # 3 125 workgroups: the LM-head grids AUTO streams up to 4 800 of (STREAM_WGS_16)
])
@pytest.mark.parametrize("dtype,out_dtype", [(ElemType.BF16, ElemType.BF16), (ElemType.F16, ElemType.F32), (ElemType.BF16, ElemType.F32)])
def test_stream64_kernel_matches_the_oracle(client, oracle, m, n, k, kw, dtype, out_dtype):
    run_case(client, oracle, m, n, k, dtype, out_dtype, True, ALGOS["stream64"], **kw)


def test_stream64_refusals_determinism_and_repeated_launches(client, oracle):
    for (m, n, k, tb) in ((65, 65, 128, True), (8, 64, 96, True), (8, 64, 128, False)):       # both extents > 64; K not a multiple of 64; row-major B
        with pytest.raises(ServerError):
            run_case(client, oracle, m, n, k, ElemType.BF16, ElemType.F32, tb, ALGOS["stream64"])
    a = TensorHandle.uniform(client, (64, 8192), ElemType.BF16, 3, 1, -1.0, 1.0)
    b = TensorHandle.uniform(client, (4096, 8192), ElemType.BF16, 3, 2, -1.0, 1.0)
    outs = []
    for _ in range(4):      # back to back on one stream: the ring, the barriers and the LDS hand-over leave nothing behind
        c = TensorHandle.new_contiguous((64, 4096), client.empty(64 * 4096 * 4), ElemType.F32)
        ops.matmul(client, a, TensorHandle.new(b.handle, (8192, 4096), (1, 8192), ElemType.BF16), c, algo=ALGOS["stream64"])
        outs.append(c.to_numpy(client).copy())
    assert all(np.array_equal(outs[0], o) for o in outs[1:])
    # the sliced form (two K slices meeting in library scratch, ticket per row block): the same bits on every launch and on another
    # stream (its own scratch and tickets), and inside a capture window on a stream that has neither -- where it falls back to the
    # unsliced form: another association, the same product
    import ctypes as C
    lib, ctx, chk = client.lib, client.ctx, client._s.check
    b2 = TensorHandle.uniform(client, (8192, 8192), ElemType.BF16, 3, 4, -1.0, 1.0)
    d = N.GemmDesc(m=64, n=8192, k=8192, batch=1, lda=8192, ldb=8192, ldc=8192, dtype_ab=N.DTYPE_BF16, dtype_c=N.DTYPE_F32, trans_b=1, algo=ALGOS["stream64"])
    outs = []
    streams = [None, C.c_void_p(), C.c_void_p()]
    for st in streams[1:]:
        chk(lib.mi355_stream_create(ctx, C.byref(st)))
    for rep, st in enumerate((None, None, streams[1])):
        c = client.empty(64 * 8192 * 4)
        chk(lib.mi355_memset(ctx, st, C.c_void_p(c.device_ptr()), 0xEE, c.size))
        chk(lib.mi355_gemm(ctx, st, C.byref(d), C.c_void_p(a.device_ptr()), C.c_void_p(b2.device_ptr()), C.c_void_p(c.device_ptr())))
        chk(lib.mi355_sync(ctx, st))
        outs.append(client.read_one(c).view(np.float32).copy())
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    c = client.empty(64 * 8192 * 4)
    g = C.c_void_p()
    chk(lib.mi355_graph_begin_capture(ctx, streams[2]))
    chk(lib.mi355_gemm(ctx, streams[2], C.byref(d), C.c_void_p(a.device_ptr()), C.c_void_p(b2.device_ptr()), C.c_void_p(c.device_ptr())))
    chk(lib.mi355_graph_end_capture(ctx, streams[2], C.byref(g)))
    chk(lib.mi355_graph_replay(ctx, streams[2], g))
    chk(lib.mi355_sync(ctx, streams[2]))
    got = client.read_one(c).view(np.float32)
    assert np.allclose(got, outs[0], rtol=0, atol=1e-3 * np.abs(outs[0]).max()) and not np.array_equal(got, outs[0])
    chk(lib.mi355_graph_destroy(ctx, g))
    for st in streams[1:]:
        chk(lib.mi355_stream_destroy(ctx, st))


# ---- ... its f32 form (gemm_stream64_f32.hip, round 5): v_mfma_f32_16x16x4_f32, both operands through per-wave LDS rings -----------
@pytest.mark.parametrize("m,n,k,kw", [
    (16, 8192, 8192, {}),                        # the shape the round-4 review names: 256 workgroups of 32 streamed rows
    (9, 1000, 2048, {}),                         # one small block with 9 valid rows; ragged streamed extent
    (16, 33, 64, {}),                            # a single K-block: three of the four waves have nothing to do
    (12, 48, 128, {"ldc": 56}),                  # two K-blocks; pitched C
    (16, 64, 192, {}),                           # three K-blocks: fewer than waves
    (17, 257, 1024, {"lda": 1032, "ldb": 1040}), # two small blocks, one valid row in the second; padded operand rows
    (32, 4096, 4096, {}),                        # two small blocks, one streamed block per workgroup
    (32, 9000, 1024, {}),                        # ... two streamed blocks per workgroup (282 workgroups), ragged
    (33, 640, 1280, {"batch": 3}),               # three small blocks
    (48, 512, 1600, {"batch": 2, "bcast_b": True}),
    (64, 2048, 2048, {}),                        # four small blocks: 160 KiB of LDS
    (64, 8200, 640, {}),                         # 257 workgroups of 32 rows: the rows-split form, ten K-blocks
    (64, 8192, 1024, {}),                        # ... its steady state (sixteen K-blocks, six stages)
    (56, 9000, 448, {}),                         # ... seven K-blocks: one steady pair, then the tail; ragged rows on both sides
    (64, 8192, 192, {}),                         # ... fewer K-blocks than stages
    (50, 8200, 64, {}),                          # ... a single K-block
    (8192, 64, 1024, {}),                        # ... few columns: roles swapped
    (8192, 16, 4096, {}),                        # N <= 64: roles swapped, output block stored transposed
    (1000, 40, 2048, {"ldc": 48}),
    (513, 10, 640, {"batch": 2}),
    (24, 300, 2048, {"batch": 40}),              # many workgroups through the batch
    (16, 28672, 4096, {}),                       # 448 MiB streamed: non-temporal pieces
])
def test_stream64_f32_form_matches_the_oracle(client, oracle, m, n, k, kw):
    run_case(client, oracle, m, n, k, ElemType.F32, ElemType.F32, True, ALGOS["stream64"], **kw)


def test_stream64_f32_form_refusals_determinism_and_capture(client, oracle):
    for (m, n, k, tb, out) in ((65, 65, 128, True, ElemType.F32), (8, 64, 96, True, ElemType.F32), (8, 64, 128, False, ElemType.F32)):
        with pytest.raises(ServerError):           # both extents > 64; K not a multiple of 64; row-major B
            run_case(client, oracle, m, n, k, ElemType.F32, out, tb, ALGOS["stream64"])
    a = TensorHandle.uniform(client, (16, 4096), ElemType.F32, 3, 1, -1.0, 1.0)
    b = TensorHandle.uniform(client, (4096, 4096), ElemType.F32, 3, 2, -1.0, 1.0)
    outs = []
    for _ in range(4):      # back to back on one stream: rings and the final fold leave nothing behind
        c = TensorHandle.new_contiguous((16, 4096), client.empty(16 * 4096 * 4), ElemType.F32)
        ops.matmul(client, a, TensorHandle.new(b.handle, (4096, 4096), (1, 4096), ElemType.F32), c, algo=ALGOS["stream64"])
        outs.append(c.to_numpy(client).copy())
    assert all(np.array_equal(outs[0], o) for o in outs[1:])
    # it needs neither scratch nor tickets: inside a capture window the same kernel runs, the same bits come out
    import ctypes as C
    lib, ctx, chk = client.lib, client.ctx, client._s.check
    d = N.GemmDesc(m=16, n=4096, k=4096, batch=1, lda=4096, ldb=4096, ldc=4096, dtype_ab=N.DTYPE_F32, dtype_c=N.DTYPE_F32, trans_b=1, algo=N.GEMM_ALGO_AUTO)
    st, g = C.c_void_p(), C.c_void_p()
    chk(lib.mi355_stream_create(ctx, C.byref(st)))
    c = client.empty(16 * 4096 * 4)
    chk(lib.mi355_graph_begin_capture(ctx, st))
    chk(lib.mi355_gemm(ctx, st, C.byref(d), C.c_void_p(a.device_ptr()), C.c_void_p(b.device_ptr()), C.c_void_p(c.device_ptr())))
    chk(lib.mi355_graph_end_capture(ctx, st, C.byref(g)))
    chk(lib.mi355_graph_replay(ctx, st, g))
    chk(lib.mi355_sync(ctx, st))
    sel = C.c_int32()
    chk(lib.mi355_gemm_select(ctx, C.byref(d), C.byref(sel)))
    if sel.value == N.GEMM_ALGO_STREAM64:
        assert np.array_equal(client.read_one(c).view(np.float32).reshape(16, 4096), outs[0])
    chk(lib.mi355_graph_destroy(ctx, g))
    chk(lib.mi355_stream_destroy(ctx, st))


@pytest.mark.parametrize("m,n,k", [(256, 256, 64), (512, 384, 128), (2048, 2048, 192), (1000, 900, 256), (2048, 2048, 320), (128, 128, 4096)])
@pytest.mark.parametrize("dtype,out_dtype", [(ElemType.BF16, ElemType.BF16), (ElemType.F16, ElemType.F32)])
def test_lp128_loader_wave_form_with_rings_shorter_than_their_depth(client, oracle, m, n, k, dtype, out_dtype):
    """At most one workgroup per CU: gemm_lp128.hip runs its 4-stage ring with four loader waves.  K of 1 ... 5 K-tiles
    exercises the prologue (fewer K-tiles than ring slots: the loaders' loop runs 0 ... 4 times) and the tail waits; the last
    shape is one workgroup walking 64 K-tiles."""
    run_case(client, oracle, m, n, k, dtype, out_dtype, True, ALGOS["lp128"])


# ---- A stored [K][M] with a row-major B: lhs^T . grad_out, the weight-gradient product (gemm_lp128.hip ATN) ---------------------
def _tn_desc(m, n, k, dtype, out, lda=None, ldb=None, batch=1, algo=N.GEMM_ALGO_AUTO, trans_b=0):
    lda, ldb = lda or m, ldb or (k if trans_b else n)
    return N.GemmDesc(m=m, n=n, k=k, batch=batch, lda=lda, ldb=ldb, ldc=n, stride_a=k * lda, stride_b=(n if trans_b else k) * ldb, stride_c=m * n,
                      dtype_ab=int(dtype), dtype_c=int(out), trans_a=1, trans_b=trans_b, algo=algo)


@pytest.mark.parametrize("dtype", [ElemType.BF16, ElemType.F16])
@pytest.mark.parametrize("m,n,k,batch,lda,out16,algo", [
    (128, 128, 64, 1, None, False, "lp128"),          # one tile, one K-tile
    (520, 776, 448, 1, 528, False, "lp128"),          # 4-stage ring + loader waves, edge tiles in both directions, padded rows of A
    (2048, 2304, 320, 2, None, True, "lp128"),        # two-stage form + loader waves, batches
    (4096, 4104, 64, 1, None, True, "lp128"),         # single-stage form
    (136, 264, 8192, 1, None, False, "lp128"),        # split-K slices
    (512, 512, 8192, 1, None, True, "auto"),          # what a weight gradient looks like: small output, long K
    (1024, 4096, 2048, 1, 1032, True, "auto"),
    (3072, 3072, 128, 1, None, True, "lp256w4"),      # the 256 x 256 kernel: one K-tile pair, every wave position
    (4104, 3080, 448, 1, 4112, False, "lp256w4"),     # ... edge tiles in both directions, padded rows of A, f32 C
    (2304, 2560, 1024, 2, None, True, "lp256w4"),     # ... batches
    (4096, 4096, 4096, 1, None, True, "auto256"),     # a weight gradient of a large layer: AUTO takes the 256 x 256 kernel natively
    (4096, 2048, 4096, 1, None, True, "auto256"),     # (late round 6) ... and where a narrow tile would run for a K-contiguous A: no scratch transposition any more
    (744, 5432, 512, 1, None, True, "auto"),          # ... the 128 x 128 kernel where the table's rows say so
    (10376, 64, 2048, 1, None, False, "auto"),        # ... 64 columns of a row-major rhs: as it is on the 128 x 128 kernel
])
def test_transposed_a_with_row_major_b_is_staged_natively_and_gives_the_bits_of_the_k_contiguous_form(client, oracle, dtype, m, n, k, batch, lda,
                                                                                                     out16, algo):
    """A [K][M] and B [K][N] are both walked along their ROWS by K: the A tile is built as the mirror image of the row-major B tile
    (blocks of [4 k][32 m], fragments through ds_read_b64_tr_b16).  Same k order inside every MFMA, same K-tile order: the bits of
    the launch with both operands K-contiguous on the same kernel; and one batch entry against the f64 oracle directly."""
    lda = lda or m
    conv, back = (oracle.to_bf16, oracle.from_bf16) if dtype == ElemType.BF16 else (oracle.to_f16, oracle.from_f16)
    a_km = conv(oracle.fill_uniform(batch * k * lda, 71, -1.0, 1.0)).reshape(batch, k, lda)
    b_kn = conv(oracle.fill_uniform(batch * k * n, 72, -1.0, 1.0)).reshape(batch, k, n)
    odt = dtype if out16 else ElemType.F32
    d = _tn_desc(m, n, k, dtype, odt, lda=lda, batch=batch)
    twin_algo = N.GEMM_ALGO_LP_256W4 if algo in ("lp256w4", "auto256") else N.GEMM_ALGO_LP_128
    if algo in ("auto", "auto256"):
        assert ops.gemm_select(client, d) == twin_algo and ops.gemm_relayout_plan(client, d) == (False, False)
        algo = "auto"
    ta, tb = TensorHandle.from_numpy(client, a_km, dtype), TensorHandle.from_numpy(client, b_kn, dtype)
    c1 = TensorHandle.new_contiguous((batch, m, n), client.empty(batch * m * n * odt.size()), odt)
    ops.matmul(client, TensorHandle.new(ta.handle, (batch, m, k), (k * lda, 1, lda), dtype), TensorHandle.new(tb.handle, (batch, k, n), (k * n, n, 1), dtype),
               c1, algo=ALGOS[algo])
    # the twin: A as [M][K], B as [N][K]
    a_mk = np.ascontiguousarray(a_km[:, :, :m].transpose(0, 2, 1))
    b_nk = np.ascontiguousarray(b_kn.transpose(0, 2, 1))
    ua, ub = TensorHandle.from_numpy(client, a_mk, dtype), TensorHandle.from_numpy(client, b_nk, dtype)
    c2 = TensorHandle.new_contiguous((batch, m, n), client.empty(batch * m * n * odt.size()), odt)
    ops.matmul(client, TensorHandle.new(ua.handle, (batch, m, k), (m * k, k, 1), dtype), TensorHandle.new(ub.handle, (batch, k, n), (k * n, 1, k), dtype),
               c2, algo=twin_algo)
    got = c1.to_numpy(client)
    assert np.array_equal(got, c2.to_numpy(client))
    bi = batch - 1
    A = back(a_mk[bi].reshape(-1)).reshape(m, k)[-96:].astype(np.float64)
    Bv = back(b_kn[bi].reshape(-1)).reshape(k, n).astype(np.float64)
    out = _decode(oracle, got.reshape(batch, m, n)[bi][-96:], odt)
    tol = REL * (np.abs(A) @ np.abs(Bv)) + (0 if not out16 else np.abs(A @ Bv) * 2.0 ** (-7 if dtype == ElemType.BF16 else -10))
    assert np.all(np.abs(out - A @ Bv) <= tol + 1e-30)


def test_transposed_a_selection_and_refusals(client, oracle):
    bf = ElemType.BF16
    # a 256-tile shape is native on the 256 x 256 kernel; where a narrower tile would run for a K-contiguous A (they stage that form only), the
    # cheaper of the two native forms (late round 6: the scratch transposition of A was never paid back by the better tile -- 42320 x 144 x 6144
    # 400.9 us re-laid out against 126.6 native, profiles/r06_random_audit_ta.txt)
    d = _tn_desc(8192, 8192, 8192, bf, bf)
    assert ops.gemm_relayout_plan(client, d) == (False, False) and ops.gemm_select(client, d) == N.GEMM_ALGO_LP_256W4
    d = _tn_desc(4096, 2048, 4096, bf, bf)
    assert ops.gemm_relayout_plan(client, d) == (False, False) and ops.gemm_select(client, d) == N.GEMM_ALGO_LP_256W4
    d = _tn_desc(744, 5432, 512, bf, bf)
    assert ops.gemm_relayout_plan(client, d) == (False, False) and ops.gemm_select(client, d) == N.GEMM_ALGO_LP_128
    # rows of C not a multiple of 8 / A and B both transposed: no native form
    assert ops.gemm_relayout_plan(client, _tn_desc(516, 512, 1024, bf, bf)) == (True, False)
    assert ops.gemm_relayout_plan(client, _tn_desc(512, 512, 1024, bf, bf, trans_b=1)) == (True, False)
    for d in (_tn_desc(516, 512, 1024, bf, bf, algo=N.GEMM_ALGO_LP_128), _tn_desc(512, 512, 1024, bf, bf, algo=N.GEMM_ALGO_LP_128, trans_b=1),
              _tn_desc(512, 512, 1024, bf, bf, algo=N.GEMM_ALGO_LP_256X128), _tn_desc(512, 512, 1024, bf, bf, algo=N.GEMM_ALGO_LP_256W4, trans_b=1),
              _tn_desc(512, 512, 1024, bf, bf, algo=N.GEMM_ALGO_LP_256P), _tn_desc(516, 512, 1024, bf, bf, algo=N.GEMM_ALGO_LP_256W4)):
        a, b, c = client.empty(2 << 20), client.empty(2 << 20), client.empty(2 << 20)
        rc = client.lib.mi355_gemm(client.ctx, None, C.byref(d), C.c_void_p(a.device_ptr()), C.c_void_p(b.device_ptr()), C.c_void_p(c.device_ptr()))
        assert rc == N.E_UNSUPPORTED
    # through AUTO those still come out right (re-layout)
    m, n, k = 516, 512, 1024
    a_km = oracle.fill_uniform(k * m, 73, -1.0, 1.0).reshape(k, m)
    b_kn = oracle.fill_uniform(k * n, 74, -1.0, 1.0).reshape(k, n)
    ta, a_val = _to_dev(client, oracle, a_km, bf)
    tb, b_val = _to_dev(client, oracle, b_kn, bf)
    c = TensorHandle.new_contiguous((m, n), client.empty(m * n * 4), ElemType.F32)
    ops.matmul(client, TensorHandle.new(ta.handle, (m, k), (1, m), bf), TensorHandle.new(tb.handle, (k, n), (n, 1), bf), c)
    A, Bv = a_val.reshape(k, m).T.astype(np.float64), b_val.reshape(k, n).astype(np.float64)
    assert np.all(np.abs(c.to_numpy(client).reshape(m, n) - A @ Bv) <= REL * (np.abs(A) @ np.abs(Bv)) + 1e-30)


def test_split_plan_on_the_device_is_what_the_launcher_does(client):
    """mi355_gemm_split_plan with this device's CU count (ops.gemm_split_plan): 512 x 512 x 8192 runs as 16 slices on 256 CUs -- and rocprofv3
    shows 16 x 16 = 256 workgroups of the split kernel plus one fold (profiles/r03_rocprof_kernel_stats_all_configs_final_tree.csv)."""
    d = N.GemmDesc(m=512, n=512, k=8192, batch=1, lda=8192, ldb=8192, ldc=512, dtype_ab=N.DTYPE_BF16, dtype_c=N.DTYPE_BF16, trans_b=1)
    assert client.properties().num_streaming_multiprocessors == 256 and ops.gemm_split_plan(client, d) == 16
    assert ops.gemm_select(client, d) == N.GEMM_ALGO_LP_128
