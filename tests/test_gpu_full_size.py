"""Parity at BASELINE.json's FULL sizes (C2 4096^3 f32, C3 8192^3 bf16, C4 1 GiB f32, C5 2048^3 batch)
through properties that do not need a full CPU recomputation, plus oracle checks on samples the CPU
finishes in seconds:

  * GEMM: exact identity (A = I returns the B operand bit for bit), exact power-of-two linearity
    (C(2A) == 2 C(A) bit for bit in f32), sampled rows against the f64-accumulating oracle (1e-5
    relative to sum|a||b|, BASELINE.json), agreement between independent kernels.
  * reduce: checksum of checksums (the full-array sum against the sum of 8 shard sums, i.e. the
    multi-GPU partition), the exact f64 oracle over the same counter-RNG data, run-to-run bit
    determinism, planted maxima / ties / NaN for the argmax rule at 2^28 elements.
"""
import numpy as np
import pytest

from cubecl_amd import ElemType, TensorHandle, ops, sharded
from cubecl_amd import _native as N

pytestmark = pytest.mark.gpu
SEED = 0x5EEDC0BE
REL = 1e-5


def _record(case, err, ref, bound, extra=None):
    """Makes the margin visible (review of round 2, weak #2): the tolerance is 1e-5 x sum|a||b| -- north_star's "1e-5 relative"
    read against the magnitude of the summands, because a dot product of 8192 mixed-sign terms cancels -- so every full-size
    check records BOTH ratios it achieved: max |err| / sum|a||b| (the asserted one) and max |err| / |ref| (the literal reading,
    over the outputs that are not themselves cancellation residue: |ref| >= 1e-3 x sum|a||b|).  One JSON line per case in
    gpurun_out/parity_margins.jsonl and on stdout (pytest -s / the captured log)."""
    import json
    import os
    from pathlib import Path
    err, ref, bound = (np.asarray(x, dtype=np.float64) for x in (err, ref, bound))
    solid = np.abs(ref) >= 1e-3 * bound
    line = {"case": case, "outputs_checked": int(err.size), "max_err_over_sum_abs_products": float((err / (bound + 1e-300)).max()),
            "max_err_over_abs_ref": float((err[solid] / np.abs(ref[solid])).max()) if solid.any() else None,
            "outputs_in_literal_reading": int(solid.sum()), "tolerance": REL}
    line.update(extra or {})
    print("PARITY_MARGIN " + json.dumps(line))
    out = Path(os.environ.get("GRAFT_REPO_ROOT", Path(__file__).resolve().parents[1])) / "gpurun_out"
    try:
        out.mkdir(exist_ok=True)
        with open(out / "parity_margins.jsonl", "a") as f:
            f.write(json.dumps(line) + "\n")
    except OSError:
        pass


def _rows_check(oracle, a_bits, b_bits, got_rows, rows, k, n, dtype, trans_b=True, case=None):
    """got_rows[i] == A[rows[i], :] . B^T (or B) within REL * sum|a||b| (f32) / one ulp (16-bit out is not used here)."""
    if dtype == ElemType.F32:
        a_val, b_val = a_bits, b_bits
    else:
        a_val = oracle.from_bf16(a_bits)
        b_val = oracle.from_bf16(b_bits)
    A = a_val.reshape(-1, k)[rows].astype(np.float64)
    Bm = b_val.reshape(n, k).astype(np.float64).T if trans_b else b_val.reshape(k, n).astype(np.float64)
    ref = A @ Bm
    bound = np.abs(A) @ np.abs(Bm)
    err = np.abs(got_rows.astype(np.float64) - ref)
    if case:
        _record(case, err, ref, bound)
    assert np.all(err <= REL * bound + 1e-30), float((err / (bound + 1e-30)).max())


def test_c3_bf16_8192_sampled_rows_linearity_and_kernel_agreement(client, oracle):
    S = 8192
    a = TensorHandle.uniform(client, (S, S), ElemType.BF16, SEED, 100, -1.0, 1.0)
    b = TensorHandle.uniform(client, (S, S), ElemType.BF16, SEED, 200, -1.0, 1.0)        # stored [N][K]
    bt = TensorHandle.new(b.handle, (S, S), (1, S), ElemType.BF16)
    c = TensorHandle.new_contiguous((S, S), client.empty(S * S * 4), ElemType.F32)
    d = N.GemmDesc(m=S, n=S, k=S, batch=1, lda=S, ldb=S, ldc=S, dtype_ab=N.DTYPE_BF16, dtype_c=N.DTYPE_F32, trans_b=1)
    assert ops.gemm_select(client, d) == N.GEMM_ALGO_LP_256M16
    ops.matmul(client, a, bt, c)
    got = c.to_numpy(client)
    rows = np.array([0, 1, 255, 256, 4095, 4096, 8191, 5003])
    a_bits = oracle.to_bf16(oracle.fill_uniform(S * S, 100, -1.0, 1.0))
    b_bits = oracle.to_bf16(oracle.fill_uniform(S * S, 200, -1.0, 1.0))
    assert np.array_equal(a.to_numpy(client).reshape(-1)[: 1 << 16], a_bits[: 1 << 16])   # device RNG == oracle RNG
    _rows_check(oracle, a_bits, b_bits, got[rows], rows, S, S, ElemType.BF16, case="C3 8192^3 bf16 -> f32 C, 8 sampled rows vs f64 oracle")
    # exact linearity under a power of two: scaling A by 2 is exact in bf16, so C doubles bit for bit
    a2 = TensorHandle.from_numpy(client, oracle.to_bf16(2.0 * oracle.from_bf16(a_bits)), ElemType.BF16)
    c2 = TensorHandle.new_contiguous((S, S), client.empty(S * S * 4), ElemType.F32)
    ops.matmul(client, TensorHandle.new(a2.handle, (S, S), (S, 1), ElemType.BF16), bt, c2)
    assert np.array_equal(c2.to_numpy(client), 2.0 * got)
    # another kernel (the 128x128 tile's loader-wave form) agrees within the f32 bound |a||b| <= 1 per product
    ops.matmul(client, a, bt, c2, algo=N.GEMM_ALGO_LP_128)
    assert np.max(np.abs(c2.to_numpy(client) - got)) <= REL * S
    # bit-reproducible from launch to launch
    ops.matmul(client, a, bt, c2)
    assert np.array_equal(c2.to_numpy(client), got)


def test_c3_bf16_8192_identity_returns_operand_bits(client, oracle):
    S = 8192
    eye = np.zeros((S, S), dtype=np.uint16)
    eye[np.arange(S), np.arange(S)] = 0x3F80                                              # bf16 1.0
    a = TensorHandle.from_numpy(client, eye, ElemType.BF16)
    b = TensorHandle.uniform(client, (S, S), ElemType.BF16, SEED, 201, -1.0, 1.0)        # [N][K]
    c = TensorHandle.new_contiguous((S, S), client.empty(S * S * 2), ElemType.BF16)
    ops.matmul(client, TensorHandle.new(a.handle, (S, S), (S, 1), ElemType.BF16),
               TensorHandle.new(b.handle, (S, S), (1, S), ElemType.BF16), c)
    # Out = I . B^T, so Out[m][n] = B[n][m]: exactly one non-zero product per output, no rounding anywhere
    assert np.array_equal(c.to_numpy(client).reshape(S, S), b.to_numpy(client).reshape(S, S).T)


def _bench_desc(m, n, k, dtype_ab, dtype_c, batch=1):
    """The descriptor bench.py builds (bench.py gemm_desc): contiguous operands, B stored [N][K], batch strides set."""
    return N.GemmDesc(m=m, n=n, k=k, batch=batch, lda=k, ldb=k, ldc=n, stride_a=m * k, stride_b=n * k, stride_c=m * n,
                      dtype_ab=dtype_ab, dtype_c=dtype_c, trans_a=0, trans_b=1, algo=N.GEMM_ALGO_AUTO)


def _bf16_rows_check(oracle, a_bits, b_bits, got_bits, rows, k, n, case=None):
    """bf16 C against the f64 oracle: |got - ref| <= one bf16 ulp of ref + 1e-5 * sum|a||b| (f32 accumulation bound
    of BASELINE.json + the single rounding of the store; reference loop runtime_tests/cmma.rs:695-722)."""
    A = oracle.from_bf16(a_bits).reshape(-1, k)[rows].astype(np.float64)
    Bm = oracle.from_bf16(b_bits).reshape(n, k).astype(np.float64).T
    ref = A @ Bm
    bound = np.abs(A) @ np.abs(Bm)
    got = oracle.from_bf16(got_bits).astype(np.float64)
    ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(ref), 2.0 ** -126))) - 7)          # bf16: 8 significant bits
    err = np.abs(got - ref)
    if case:
        _record(case, err, ref, bound, {"note": "bf16 C: the error includes the one rounding of the store (<= 2^-8 relative), allowed as one ulp",
                                        "max_err_over_allowed": float((err / (ulp + REL * bound)).max())})
    assert np.all(err <= ulp + REL * bound), float((err / (ulp + REL * bound)).max())
    # and the rounding is to nearest: at least ~99 % of the outputs are the correctly rounded f64 result
    exact = oracle.to_bf16(ref.astype(np.float32))
    assert np.mean(exact == got_bits) > 0.98


def test_c3_bf16_8192_bf16_output_as_benched(client, oracle):
    """Config C3 in the EXACT form bench.py times it (bench.py:228-238): 8192^3 bf16, f32 accumulate, **bf16 C**, the
    bench's descriptor through mi355_gemm with AUTO selection -- sampled rows against the f64-accumulating oracle."""
    import ctypes as C
    S = 8192
    a = TensorHandle.uniform(client, (S, S), ElemType.BF16, SEED, 100, -1.0, 1.0)
    b = TensorHandle.uniform(client, (S, S), ElemType.BF16, SEED, 200, -1.0, 1.0)        # stored [N][K]
    c = TensorHandle.new_contiguous((S, S), client.empty(S * S * 2), ElemType.BF16)
    d = _bench_desc(S, S, S, N.DTYPE_BF16, N.DTYPE_BF16)
    assert ops.gemm_select(client, d) == N.GEMM_ALGO_LP_256QM            # round 6: the persistent dripped-store loop on 16x16x32 MFMAs
    client._s.check(client.lib.mi355_gemm(client.ctx, None, C.byref(d), C.c_void_p(a.device_ptr()), C.c_void_p(b.device_ptr()),
                                          C.c_void_p(c.device_ptr())))
    got = c.to_numpy(client).reshape(S, S)
    rows = np.array([0, 1, 15, 16, 127, 128, 255, 256, 4095, 4096, 8191, 5003, 6144 + 77])      # every wave row-block position
    a_bits = oracle.to_bf16(oracle.fill_uniform(S * S, 100, -1.0, 1.0))
    b_bits = oracle.to_bf16(oracle.fill_uniform(S * S, 200, -1.0, 1.0))
    assert np.array_equal(b.to_numpy(client).reshape(-1)[-(1 << 16):], b_bits[-(1 << 16):])   # device RNG == oracle RNG
    _bf16_rows_check(oracle, a_bits, b_bits, got[rows], rows, S, S, case="C3 8192^3 bf16 -> bf16 C as benched, 11 sampled rows vs f64 oracle")
    # the f32-output form of the same launch rounds to the same bf16 values (one rounding, after the f32 accumulation) -- over
    # ALL 8192 rows, i.e. every one of the 1024 tiles of the benched <bf16 C> instantiation, the XCD-remapped ones included;
    # the f32-C instantiation itself is held to the f64 oracle and to an independent kernel over the full output above
    c32 = TensorHandle.new_contiguous((S, S), client.empty(S * S * 4), ElemType.F32)
    ops.matmul(client, a, TensorHandle.new(b.handle, (S, S), (1, S), ElemType.BF16), c32)
    full32 = c32.to_numpy(client).reshape(S, S)
    assert np.array_equal(oracle.to_bf16(full32).reshape(S, S), got)


def test_mid_size_4096x2048x4096_as_benched_takes_the_192x192_tile(client, oracle):
    """bench.py's `4096x2048x4096` entry: AUTO takes the 192 x 192 tile of the 4-wave kernel since round 5 (242 tiles, one round;
    until then the 256 x 128 form of the mid-size kernel); all outputs equal the 128x128 kernel's bit for bit (same k order per
    output) and sampled rows meet the f64 oracle."""
    import ctypes as C
    m, n, k = 4096, 2048, 4096
    a = TensorHandle.uniform(client, (m, k), ElemType.BF16, SEED, 700, -1.0, 1.0)
    b = TensorHandle.uniform(client, (n, k), ElemType.BF16, SEED, 701, -1.0, 1.0)
    d = _bench_desc(m, n, k, N.DTYPE_BF16, N.DTYPE_BF16)
    assert ops.gemm_select(client, d) == N.GEMM_ALGO_LP_192X192
    outs = []
    for algo in (N.GEMM_ALGO_AUTO, N.GEMM_ALGO_LP_128):
        d.algo = algo
        c = TensorHandle.new_contiguous((m, n), client.empty(m * n * 2), ElemType.BF16)
        client._s.check(client.lib.mi355_gemm(client.ctx, None, C.byref(d), C.c_void_p(a.device_ptr()), C.c_void_p(b.device_ptr()),
                                              C.c_void_p(c.device_ptr())))
        outs.append(c.to_numpy(client).reshape(m, n))
    assert np.array_equal(outs[0], outs[1])
    rows = np.array([0, 127, 128, 255, 256, 2047, 2048 + 129, 4095])
    a_bits = oracle.to_bf16(oracle.fill_uniform(m * k, 700, -1.0, 1.0))
    b_bits = oracle.to_bf16(oracle.fill_uniform(n * k, 701, -1.0, 1.0))
    _bf16_rows_check(oracle, a_bits, b_bits, outs[0][rows], rows, k, n, case="4096x2048x4096 bf16 on the 192x192 tile, 8 sampled rows vs f64 oracle")


def _nn_bench_desc(m, n, k, dtype_ab, dtype_c, batch=1):
    """bench.py's descriptor for the reference's default rhs layout: B row-major [K][N] (trans_b = 0)."""
    return N.GemmDesc(m=m, n=n, k=k, batch=batch, lda=k, ldb=n, ldc=n, stride_a=m * k, stride_b=n * k, stride_c=m * n,
                      dtype_ab=dtype_ab, dtype_c=dtype_c, trans_a=0, trans_b=0, algo=N.GEMM_ALGO_AUTO)


def test_c3_bf16_8192_row_major_rhs_is_native_and_bit_identical_to_the_k_contiguous_form(client, oracle):
    """Config C3 with the rhs as TensorHandle::new_contiguous lays it out ([K][N], crates/cubecl-std/src/tensor/handle.rs:89):
    no operand is copied (empty re-layout plan, no library scratch), the 256x256 kernel stages B through its transposing-read
    image (ds_read_b64_tr_b16 of a half-swapped block image, gemm_lp256qm.hip), and all 64 Mi outputs equal those of the K-contiguous launch (held to the f64 oracle above) bit for bit."""
    import ctypes as C
    S = 8192
    a = TensorHandle.uniform(client, (S, S), ElemType.BF16, SEED, 100, -1.0, 1.0)
    b_nk = TensorHandle.uniform(client, (S, S), ElemType.BF16, SEED, 200, -1.0, 1.0)      # [N][K]
    b_kn = ops.into_contiguous(client, TensorHandle.new(b_nk.handle, (S, S), (1, S), ElemType.BF16))   # its transpose, [K][N]
    outs = []
    for d, b in ((_bench_desc(S, S, S, N.DTYPE_BF16, N.DTYPE_BF16), b_nk), (_nn_bench_desc(S, S, S, N.DTYPE_BF16, N.DTYPE_BF16), b_kn)):
        # (AUTO: the persistent 16x16x32 kernel for both layouts since the second K loop of round 6 -- its transposing-read form runs the
        #  same MFMA chains as its [N][K] form, so the pair is compared as AUTO launches it)
        assert ops.gemm_select(client, d) == N.GEMM_ALGO_LP_256QM and ops.gemm_relayout_plan(client, d) == (False, False)
        c = TensorHandle.new_contiguous((S, S), client.empty(S * S * 2), ElemType.BF16)
        client._s.check(client.lib.mi355_gemm(client.ctx, None, C.byref(d), C.c_void_p(a.device_ptr()), C.c_void_p(b.device_ptr()),
                                              C.c_void_p(c.device_ptr())))
        outs.append(c.to_numpy(client))
    assert np.array_equal(outs[0], outs[1])
    # and directly against the oracle on a few rows (the transposition above is a device kernel too)
    rows = np.array([3, 255, 4097, 8190])
    a_bits = oracle.to_bf16(oracle.fill_uniform(S * S, 100, -1.0, 1.0))
    b_bits = oracle.to_bf16(oracle.fill_uniform(S * S, 200, -1.0, 1.0))
    _bf16_rows_check(oracle, a_bits, b_bits, outs[1].reshape(S, S)[rows], rows, S, S, case="C3 8192^3 bf16, row-major rhs (NN), 4 sampled rows vs f64 oracle")


def test_c5_batch64_2048_bf16_row_major_rhs_is_native(client, oracle):
    """Config C5's shard with row-major rhs matrices: native (no re-layout), bit-identical to the K-contiguous launch."""
    import ctypes as C
    B, M = 64, 2048
    a = TensorHandle.uniform(client, (B, M, M), ElemType.BF16, SEED, 500, -1.0, 1.0)
    b_nk = TensorHandle.uniform(client, (B, M, M), ElemType.BF16, SEED, 600, -1.0, 1.0)
    b_kn = ops.into_contiguous(client, TensorHandle.new(b_nk.handle, (B, M, M), (M * M, 1, M), ElemType.BF16))
    outs = []
    for d, b in ((_bench_desc(M, M, M, N.DTYPE_BF16, N.DTYPE_BF16, batch=B), b_nk), (_nn_bench_desc(M, M, M, N.DTYPE_BF16, N.DTYPE_BF16, batch=B), b_kn)):
        assert ops.gemm_relayout_plan(client, d) == (False, False)
        # the dripped-store kernels: 16x16x32 form for the K-contiguous rhs since round 6 (another summation order), 32x32x16 form for the
        # row-major rhs; the statement here is about the STAGING of the 32x32x16 kernel, so that kernel is named for the K-contiguous launch
        assert ops.gemm_select(client, d) == N.GEMM_ALGO_LP_256QM
        d.algo = N.GEMM_ALGO_LP_256Q
        c = TensorHandle.new_contiguous((B, M, M), client.empty(B * M * M * 2), ElemType.BF16)
        client._s.check(client.lib.mi355_gemm(client.ctx, None, C.byref(d), C.c_void_p(a.device_ptr()), C.c_void_p(b.device_ptr()),
                                              C.c_void_p(c.device_ptr())))
        outs.append(c.to_numpy(client))
    assert np.array_equal(outs[0], outs[1])


def test_c5_batch64_2048_bf16_as_benched_takes_the_dripped_store_kernel(client, oracle):
    """Config C5's per-GPU shard of an 8-GPU job (bench.py batched_c5, `shard_of_64`): batch 64 of 2048^3 bf16 -> bf16 C.  (The
    job bench.py times at N = 1 is the whole batch of 512: test_c5_batch512_2048_bf16_as_benched below.)
    AUTO must take the persistent 256x256 kernel; sampled rows of five matrices (first, last, three inside) against the
    f64 oracle, operands regenerated window by window from the counter RNG."""
    import ctypes as C
    B, M = 64, 2048
    a = TensorHandle.uniform(client, (B, M, M), ElemType.BF16, SEED, 500, -1.0, 1.0)
    b = TensorHandle.uniform(client, (B, M, M), ElemType.BF16, SEED, 600, -1.0, 1.0)
    c = TensorHandle.new_contiguous((B, M, M), client.empty(B * M * M * 2), ElemType.BF16)
    d = _bench_desc(M, M, M, N.DTYPE_BF16, N.DTYPE_BF16, batch=B)
    assert ops.gemm_select(client, d) == N.GEMM_ALGO_LP_256QM      # the persistent kernel with dripped stores on 16x16x32 MFMAs (gemm_lp256qm.hip)
    client._s.check(client.lib.mi355_gemm(client.ctx, None, C.byref(d), C.c_void_p(a.device_ptr()), C.c_void_p(b.device_ptr()),
                                          C.c_void_p(c.device_ptr())))
    rows = np.array([0, 95, 96, 127, 128, 255, 256, 1023, 1024 + 129, 2047])   # held row blocks and the boundary block of a wave
    mm = M * M
    for bi in (0, 1, 31, 40, 63):
        a_bits = oracle.to_bf16(oracle.fill_uniform_at(bi * mm, mm, 500, -1.0, 1.0))
        b_bits = oracle.to_bf16(oracle.fill_uniform_at(bi * mm, mm, 600, -1.0, 1.0))
        got = client.read_one(c.handle.offset_start_by(2 * bi * mm).offset_end_by(2 * (B - 1 - bi) * mm)).view(np.uint16).reshape(M, M)
        if bi in (0, 63):                                                              # the device operand IS the oracle's
            dev = client.read_one(a.handle.offset_start_by(2 * bi * mm).offset_end_by(2 * (B - 1 - bi) * mm)).view(np.uint16)
            assert np.array_equal(dev, a_bits)
        _bf16_rows_check(oracle, a_bits, b_bits, got[rows], rows, M, M, case=f"C5 shard 64 x 2048^3 bf16 -> bf16 C, matrix {bi}, 10 sampled rows")
    # the one-tile-per-workgroup kernel on the same MFMA shape computes the same tiles bit for bit (same per-tile summation order)
    c2 = TensorHandle.new_contiguous((B, M, M), client.empty(B * M * M * 2), ElemType.BF16)
    d.algo = N.GEMM_ALGO_LP_256M16
    client._s.check(client.lib.mi355_gemm(client.ctx, None, C.byref(d), C.c_void_p(a.device_ptr()), C.c_void_p(b.device_ptr()),
                                          C.c_void_p(c2.device_ptr())))
    for bi in (0, 17, 63):
        w = lambda t: client.read_one(t.handle.offset_start_by(2 * bi * mm).offset_end_by(2 * (B - 1 - bi) * mm))
        assert np.array_equal(w(c), w(c2))


@pytest.mark.parametrize("layout", ["NT", "NN"])
def test_c5_batch512_2048_bf16_as_benched(client, oracle, layout):
    """Config C5 in the EXACT form bench.py times it at N = 1 (bench.py batched_c5, `batched_gemm_2048_bf16` and its
    `row_major_rhs_NN`): the whole batch of 512 x 2048^3 bf16 -> bf16 C on one GPU -- 3 x 4 GiB, so matrix 256 starts at byte
    2^31 of every operand (element 2^30: past what a signed 32-bit byte offset reaches) and the last matrix ends at byte 2^32.
    AUTO must take the persistent kernel
    with dripped stores (gemm_lp256q.hip); sampled rows of matrices 0, 255, 256, 300 and 511 against the f64 oracle (operands
    regenerated window by window from the counter RNG; reference loop crates/cubecl-core/src/runtime_tests/cmma.rs:695-722,
    SURVEY.md 8(d) C5), and the one-tile-per-workgroup kernel (gemm_lp256w4.hip, launched on single matrices through 64-bit
    pointer offsets) computes matrices 256 and 511 bit for bit."""
    import ctypes as C
    B, M = 512, 2048
    mm = M * M
    nn = layout == "NN"
    a = TensorHandle.uniform(client, (B, M, M), ElemType.BF16, SEED, 500, -1.0, 1.0)
    b = TensorHandle.uniform(client, (B, M, M), ElemType.BF16, SEED, 600, -1.0, 1.0)       # NT: [N][K]; NN: the same bytes read as [K][N]
    c = TensorHandle.new_contiguous((B, M, M), client.empty(B * mm * 2), ElemType.BF16)
    assert 2 * 256 * mm == 1 << 31 and 2 * B * mm == 1 << 32
    client._s.check(client.lib.mi355_memset(client.ctx, None, C.c_void_p(c.device_ptr()), 0xEE, B * mm * 2))
    d = (_nn_bench_desc if nn else _bench_desc)(M, M, M, N.DTYPE_BF16, N.DTYPE_BF16, batch=B)
    assert ops.gemm_select(client, d) == N.GEMM_ALGO_LP_256QM
    assert ops.gemm_relayout_plan(client, d) == (False, False)
    client._s.check(client.lib.mi355_gemm(client.ctx, None, C.byref(d), C.c_void_p(a.device_ptr()), C.c_void_p(b.device_ptr()),
                                          C.c_void_p(c.device_ptr())))
    win = lambda t, bi: client.read_one(t.handle.offset_start_by(2 * bi * mm).offset_end_by(2 * (B - 1 - bi) * mm)).view(np.uint16)
    rows = np.array([0, 95, 96, 127, 128, 255, 256, 1023, 1024 + 129, 2047])
    for bi in (0, 255, 256, 300, 511):
        a_bits = oracle.to_bf16(oracle.fill_uniform_at(bi * mm, mm, 500, -1.0, 1.0))
        b_bits = oracle.to_bf16(oracle.fill_uniform_at(bi * mm, mm, 600, -1.0, 1.0))
        if bi in (256, 511):                                                          # the device operands ARE the oracle's, past 4 GiB too
            assert np.array_equal(win(a, bi), a_bits) and np.array_equal(win(b, bi), b_bits)
        got = win(c, bi).reshape(M, M)
        assert not np.any(got == 0xEEEE)                                                # every output of the matrix was written
        if nn:                                                                          # [K][N] -> the [N][K] form the row check takes
            b_bits = np.ascontiguousarray(b_bits.reshape(M, M).T).reshape(-1)
        _bf16_rows_check(oracle, a_bits, b_bits, got[rows], rows, M, M,
                         case=f"C5 as benched: 512 x 2048^3 bf16 -> bf16 C ({layout}), matrix {bi}, 10 sampled rows vs f64 oracle")
    # the one-tile-per-workgroup kernel on single matrices of the upper half: same per-tile summation order => same bits
    d1 = (_nn_bench_desc if nn else _bench_desc)(M, M, M, N.DTYPE_BF16, N.DTYPE_BF16, batch=1)
    # (NT: the one-tile-per-workgroup kernel with the persistent form's MFMA shape; NN: that kernel has no row-major form -- the same kernel on the
    #  single matrix, another grid and tile walk, reached through 64-bit pointer offsets)
    d1.algo = N.GEMM_ALGO_LP_256QM if nn else N.GEMM_ALGO_LP_256M16
    c1 = TensorHandle.new_contiguous((M, M), client.empty(mm * 2), ElemType.BF16)
    for bi in (256, 511):
        off = 2 * bi * mm
        client._s.check(client.lib.mi355_gemm(client.ctx, None, C.byref(d1), C.c_void_p(a.device_ptr() + off),
                                              C.c_void_p(b.device_ptr() + off), C.c_void_p(c1.device_ptr())))
        assert np.array_equal(c1.to_numpy(client).reshape(-1), win(c, bi))
    del a, b, c, c1
    client.memory_cleanup()


@pytest.mark.parametrize("config", ["C3", "C2"])
def test_reference_arithmetic_margin(client, oracle, config):
    """north_star: "within 1e-5 relative".  The tests above read that against sum|a||b| because a dot product of thousands of
    mixed-sign terms cancels; this test turns the argument into evidence (review of round 3, weak #2).  For sampled rows of
    C3 (8192^3 bf16 -> f32 C) and C2 (4096^3 f32) it runs the reference's OWN arithmetic -- oracle_gemm(acc_f64 = 0): operands
    widened to f32, `sum += l * r` sequentially over k with separate multiply and add, the loop of
    crates/cubecl-core/src/runtime_tests/cmma.rs:695-722 -- beside the f64 oracle and records, per config, the literal
    max |x - f64| / |f64| of (a) the device and (b) that restatement.  bf16 (C3): the device's blocked MFMA accumulation (16 exact
    products per step) must be at least as close to the exact product as the reference loop is.  f32 (C2): v_mfma_f32_32x32x2_f32
    adds two products per step, so the device sums nearly as sequentially as the loop does and the two err alike (measured: rms
    2.449e-5 against 2.444e-5, literal maxima 7.3e-5 / 5.8e-5) -- asserted within 25 %.  Either way the reference's own arithmetic
    misses a literal 1e-5 by the same factor the device does: the tolerance can only be read against sum|a||b|."""
    import ctypes as C
    if config == "C3":
        S, dt, et, sa, sb = 8192, N.DTYPE_BF16, ElemType.BF16, 100, 200
    else:
        S, dt, et, sa, sb = 4096, N.DTYPE_F32, ElemType.F32, 400, 401
    a = TensorHandle.uniform(client, (S, S), et, SEED, sa, -1.0, 1.0)
    b = TensorHandle.uniform(client, (S, S), et, SEED, sb, -1.0, 1.0)                     # [N][K]
    c = TensorHandle.new_contiguous((S, S), client.empty(S * S * 4), ElemType.F32)
    ops.matmul(client, a, TensorHandle.new(b.handle, (S, S), (1, S), et), c)
    rows = np.array([5003 % S, 0, S - 1, 257])
    got = c.to_numpy(client).reshape(S, S)[rows].astype(np.float64)
    a_h, b_h = oracle.fill_uniform(S * S, sa, -1.0, 1.0), oracle.fill_uniform(S * S, sb, -1.0, 1.0)
    if config == "C3":
        a_h, b_h = oracle.to_bf16(a_h), oracle.to_bf16(b_h)
        a_val, b_val = oracle.from_bf16(a_h), oracle.from_bf16(b_h)
        odt = oracle.DT_BF16
    else:
        a_val, b_val = a_h, b_h
        odt = oracle.DT_F32
    a_rows = np.ascontiguousarray(a_h.reshape(S, S)[rows])
    ref32 = oracle.gemm(a_rows, b_h, len(rows), S, S, dtype_ab=odt, dtype_c=oracle.DT_F32, trans_b=True).reshape(len(rows), S).astype(np.float64)
    A = a_val.reshape(S, S)[rows].astype(np.float64)
    Bm = b_val.reshape(S, S).astype(np.float64).T
    ref64, bound = A @ Bm, np.abs(A) @ np.abs(Bm)
    e_dev, e_ref = np.abs(got - ref64), np.abs(ref32 - ref64)
    solid = np.abs(ref64) >= 1e-3 * bound
    lit = lambda e: float((e[solid] / np.abs(ref64[solid])).max())
    rel = lambda e: float((e / bound).max())
    _record(f"{config}: device vs the reference's own f32 sequential loop (oracle_gemm acc_f64=0), {len(rows)} rows x {S} columns", e_dev, ref64, bound, {
        "device_max_err_over_abs_ref": lit(e_dev), "reference_loop_max_err_over_abs_ref": lit(e_ref),
        "device_max_err_over_sum_abs_products": rel(e_dev), "reference_loop_max_err_over_sum_abs_products": rel(e_ref),
        "device_rms_err": float(np.sqrt(np.mean(e_dev ** 2))), "reference_loop_rms_err": float(np.sqrt(np.mean(e_ref ** 2))),
        "note": "the reference's cmma.rs:695-722 arithmetic itself misses a literal 1e-5 relative by the factor the device does"})
    slack = 1.0 if config == "C3" else 1.25
    assert rel(e_dev) <= REL
    assert rel(e_dev) <= slack * rel(e_ref)
    assert lit(e_dev) <= 1.5 * lit(e_ref)          # (a maximum of ratios over near-cancelling outputs: recorded exactly, asserted with slack)
    assert np.sqrt(np.mean(e_dev ** 2)) <= slack * np.sqrt(np.mean(e_ref ** 2))
    assert lit(e_ref) > REL                        # the reference loop itself is outside a literal 1e-5


def test_c4_one_gib_fused_sum_argmax_equals_separate_passes_and_oracle(client, oracle):
    """The fused pass the multi-GPU path launches (mi355_sum_argmax_f32, bench.py reduce_c4 / sharded.py) on the full
    1 GiB array: bit-identical to the separate sum and argmax kernels, and equal to the CPU oracle."""
    n = 1 << 28
    x = TensorHandle.uniform(client, (n,), ElemType.F32, SEED, 300, 0.0, 1.0)
    s1 = TensorHandle.new_contiguous((1,), client.empty(8), ElemType.F32)
    s2 = TensorHandle.new_contiguous((1,), client.empty(8), ElemType.F32)
    i1 = TensorHandle.new_contiguous((1,), client.empty(8), ElemType.U64)
    i2 = TensorHandle.new_contiguous((1,), client.empty(8), ElemType.U64)
    v1 = TensorHandle.new_contiguous((1,), client.empty(8), ElemType.F32)
    v2 = TensorHandle.new_contiguous((1,), client.empty(8), ElemType.F32)
    ops.reduce_sum(client, x, s1)
    ops.argmax(client, x, i1, v1)
    ops.sum_argmax(client, x, s2, i2, v2)
    assert np.array_equal(s1.to_numpy(client).view(np.uint32), s2.to_numpy(client).view(np.uint32))   # same tree, same bits
    assert int(i1.to_numpy(client)[0]) == int(i2.to_numpy(client)[0])
    assert np.array_equal(v1.to_numpy(client).view(np.uint32), v2.to_numpy(client).view(np.uint32))
    host = oracle.fill_uniform(n, 300, 0.0, 1.0)
    exact = oracle.sum_f64(host)
    assert abs(float(s2.to_numpy(client)[0]) - exact) <= REL * exact
    o_idx, o_val = oracle.argmax(host)
    assert int(i2.to_numpy(client)[0]) == o_idx and float(v2.to_numpy(client)[0]) == float(o_val)     # bit-exact index
    # run-to-run determinism of the fused pass
    ops.sum_argmax(client, x, s1, i1, v1)
    assert np.array_equal(s1.to_numpy(client).view(np.uint32), s2.to_numpy(client).view(np.uint32))
    assert int(i1.to_numpy(client)[0]) == int(i2.to_numpy(client)[0])
    # every 1/8 slice (what each rank of the 8-GPU job reduces) through the fused pass, combined as sharded.py does
    parts, pairs = [], []
    for r in range(8):
        start, count = sharded.shard_aligned_range(n, r, 8, 4)
        view = TensorHandle.new_contiguous((count,), x.handle.offset_start_by(4 * start).offset_end_by(4 * (n - start - count)),
                                           ElemType.F32)
        ops.sum_argmax(client, view, s1, i1, v1)
        parts.append(float(s1.to_numpy(client)[0]))
        pairs.append((float(v1.to_numpy(client)[0]), start + int(i1.to_numpy(client)[0])))
    assert abs(sum(parts) - exact) <= REL * exact
    assert sharded.combine_argmax(pairs) == (float(o_val), o_idx)


def test_c4_one_gib_max_min_mean_argmin(client, oracle):
    """The reduce operations added in round 4 at config C4's size (1 GiB of f32): max / min bit for bit, mean within 1e-5, argmin
    bit-exact against the oracle, with a planted minimum duplicated in two shards of the 8-way partition (the lower index wins)."""
    n = 1 << 28
    x = TensorHandle.uniform(client, (n,), ElemType.F32, SEED, 300, 0.0, 1.0)
    host = oracle.fill_uniform(n, 300, 0.0, 1.0)
    out = TensorHandle.new_contiguous((1,), client.empty(8), ElemType.F32)
    idx = TensorHandle.new_contiguous((1,), client.empty(8), ElemType.U64)
    val = TensorHandle.new_contiguous((1,), client.empty(8), ElemType.F32)
    for op in ("max", "min"):
        ops.reduce(client, x, out, op)
        assert out.to_numpy(client).view(np.uint32)[0] == np.float32(oracle.reduce_value(host, op)).view(np.uint32)
    ops.reduce(client, x, out, "mean")
    exact = oracle.sum_f64(host) / n
    assert abs(float(out.to_numpy(client)[0]) - exact) <= REL * exact
    ops.argmin(client, x, idx, val)
    ri, rv = oracle.argmin(host)
    assert int(idx.to_numpy(client)[0]) == ri and float(val.to_numpy(client)[0]) == float(rv)
    for i in (200_000_003, 17_000_001):
        client.write(x.handle.offset_start_by(4 * i).offset_end_by(4 * (n - i - 1)), np.array([-2.5], dtype=np.float32))
    ops.argmin(client, x, idx, val)
    assert int(idx.to_numpy(client)[0]) == 17_000_001 and float(val.to_numpy(client)[0]) == -2.5
    ops.reduce(client, x, out, "min")
    assert float(out.to_numpy(client)[0]) == -2.5


def test_c2_f32_4096_sampled_rows_both_layouts(client, oracle):
    M = 4096
    a = TensorHandle.uniform(client, (M, M), ElemType.F32, SEED, 400, -1.0, 1.0)
    b = TensorHandle.uniform(client, (M, M), ElemType.F32, SEED, 401, -1.0, 1.0)
    a_host = oracle.fill_uniform(M * M, 400, -1.0, 1.0)
    b_host = oracle.fill_uniform(M * M, 401, -1.0, 1.0)
    rows = np.array([0, 31, 32, 2047, 2048, 4095])
    c = TensorHandle.new_contiguous((M, M), client.empty(M * M * 4), ElemType.F32)
    for trans_b in (True, False):
        bt = TensorHandle.new(b.handle, (M, M), (1, M) if trans_b else (M, 1), ElemType.F32)
        ops.matmul(client, a, bt, c)
        got = c.to_numpy(client)
        _rows_check(oracle, a_host, b_host, got[rows], rows, M, M, ElemType.F32, trans_b=trans_b,
                    case=f"C2 4096^3 f32, {'B stored [N][K]' if trans_b else 'row-major B'}, 6 sampled rows vs f64 oracle")
        ref = TensorHandle.new_contiguous((M, M), client.empty(M * M * 4), ElemType.F32)
        ops.matmul(client, a, bt, ref, algo=N.GEMM_ALGO_F32_MFMA)                        # the 128x128 kernel
        assert np.max(np.abs(ref.to_numpy(client) - got)) <= REL * M


def test_c5_batched_2048_bf16_shard(client, oracle):
    B, M = 8, 2048            # one GPU's shard is 64 matrices; 8 keep the host check short, same launch path (grid.y = batch)
    a = TensorHandle.uniform(client, (B, M, M), ElemType.BF16, SEED, 500, -1.0, 1.0)
    b = TensorHandle.uniform(client, (B, M, M), ElemType.BF16, SEED, 600, -1.0, 1.0)
    c = TensorHandle.new_contiguous((B, M, M), client.empty(B * M * M * 4), ElemType.F32)
    ops.matmul(client, a, TensorHandle.new(b.handle, (B, M, M), (M * M, 1, M), ElemType.BF16), c)
    got = c.to_numpy(client).reshape(B, M, M)
    a_bits = oracle.to_bf16(oracle.fill_uniform(B * M * M, 500, -1.0, 1.0)).reshape(B, M * M)
    b_bits = oracle.to_bf16(oracle.fill_uniform(B * M * M, 600, -1.0, 1.0)).reshape(B, M * M)
    rows = np.array([0, 1023, 2047])
    for bi in (0, 3, 7):
        _rows_check(oracle, a_bits[bi], b_bits[bi], got[bi][rows], rows, M, M, ElemType.BF16)
    # shards of the batch are contiguous runs that tile it exactly once (what bench.py --gpus N launches per rank)
    spans = [sharded.shard_range(512, r, 8) for r in range(8)]
    assert spans == [(64 * r, 64) for r in range(8)]


def test_c4_one_gib_sum_checksum_of_checksums_and_oracle(client, oracle):
    n = 1 << 28
    x = TensorHandle.uniform(client, (n,), ElemType.F32, SEED, 300, 0.0, 1.0)
    out = TensorHandle.new_contiguous((1,), client.empty(8), ElemType.F32)
    ops.reduce_sum(client, x, out)
    total = out.to_numpy(client).copy()
    ops.reduce_sum(client, x, out)
    assert np.array_equal(out.to_numpy(client).view(np.uint32), total.view(np.uint32))   # deterministic tree
    # checksum of checksums: the 8-GPU partition of config C4, each shard reduced on its own
    parts = []
    for r in range(8):
        start, count = sharded.shard_aligned_range(n, r, 8, 4)
        view = TensorHandle.new_contiguous((count,), x.handle.offset_start_by(4 * start).offset_end_by(4 * (n - start - count)),
                                           ElemType.F32)
        ops.reduce_sum(client, view, out)
        parts.append(float(out.to_numpy(client)[0]))
    assert abs(sum(parts) - float(total[0])) <= REL * float(total[0])
    # the exact value: f64 sum of the identical counter-RNG stream on the CPU (seconds)
    exact = oracle.sum_f64(oracle.fill_uniform(n, 300, 0.0, 1.0))
    assert abs(float(total[0]) - exact) <= REL * exact
    assert abs(sum(parts) - exact) <= REL * exact


def test_c4_one_gib_argmax_planted_maxima(client, oracle):
    n = 1 << 28
    x = TensorHandle.uniform(client, (n,), ElemType.F32, SEED, 301, 0.0, 1.0)
    idx = TensorHandle.new_contiguous((1,), client.empty(8), ElemType.U64)
    val = TensorHandle.new_contiguous((1,), client.empty(8), ElemType.F32)
    s = TensorHandle.new_contiguous((1,), client.empty(8), ElemType.F32)

    def poke(i, v):
        client.write(x.handle.offset_start_by(4 * i).offset_end_by(4 * (n - i - 1)), np.array([v], dtype=np.float32))

    # a maximum planted twice, in different shards of the 8-way partition: the lower index wins
    poke(200_000_003, 2.5)
    poke(17_000_001, 2.5)
    ops.argmax(client, x, idx, val)
    assert int(idx.to_numpy(client)[0]) == 17_000_001 and float(val.to_numpy(client)[0]) == 2.5
    ops.sum_argmax(client, x, s, idx, val)                                                # fused pass agrees
    assert int(idx.to_numpy(client)[0]) == 17_000_001
    # the multi-GPU combine over per-shard winners gives the same answer
    pairs = []
    for r in range(8):
        start, count = sharded.shard_aligned_range(n, r, 8, 4)
        view = TensorHandle.new_contiguous((count,), x.handle.offset_start_by(4 * start).offset_end_by(4 * (n - start - count)),
                                           ElemType.F32)
        ops.argmax(client, view, idx, val)
        pairs.append((float(val.to_numpy(client)[0]), start + int(idx.to_numpy(client)[0])))
    assert sharded.combine_argmax(pairs) == (2.5, 17_000_001)
    # NaN ranks above everything, first NaN wins; the last element is reachable
    poke(n - 1, np.float32("nan"))
    ops.argmax(client, x, idx, val)
    assert int(idx.to_numpy(client)[0]) == n - 1 and np.isnan(val.to_numpy(client)[0])
    poke(123, np.float32("nan"))
    ops.argmax(client, x, idx, val)
    assert int(idx.to_numpy(client)[0]) == 123


def test_reductions_beyond_2_pow_32_elements(client):
    """Maximum sizes: indices that do not fit 32 bits (the output index is u64) and a ragged tail."""
    n = (1 << 32) + 4099                                   # 16 GiB + 16 396 B of f32
    h = client.empty(4 * n)
    client._s.check(client.lib.mi355_memset(client.ctx, None, h.device_ptr(), 0, 4 * n))
    x = TensorHandle.new_contiguous((n,), h, ElemType.F32)
    idx = TensorHandle.new_contiguous((1,), client.empty(8), ElemType.U64)
    val = TensorHandle.new_contiguous((1,), client.empty(8), ElemType.F32)
    s = TensorHandle.new_contiguous((1,), client.empty(8), ElemType.F32)

    def poke(i, v):
        client.write(h.offset_start_by(4 * i).offset_end_by(4 * (n - i - 1)), np.array([v], dtype=np.float32))
    planted = {5: 0.5, (1 << 31) + 3: 2.0, (1 << 32) + 17: 4.0, n - 1: 4.0, (1 << 32) - 1: -8.0}
    for i, v in planted.items():
        poke(i, v)
    ops.argmax(client, x, idx, val)
    assert int(idx.to_numpy(client)[0]) == (1 << 32) + 17 and float(val.to_numpy(client)[0]) == 4.0      # first of the two 4.0
    ops.reduce_sum(client, x, s)
    assert float(s.to_numpy(client)[0]) == sum(planted.values())                                         # exact: everything else is 0
    ops.sum_argmax(client, x, s, idx, val)
    assert int(idx.to_numpy(client)[0]) == (1 << 32) + 17 and float(s.to_numpy(client)[0]) == sum(planted.values())
    poke((1 << 32) + 17, 0.0)
    ops.argmax(client, x, idx, val)
    assert int(idx.to_numpy(client)[0]) == n - 1                                                         # the very last element
    del x, h
    client.memory_cleanup()


def test_gemm_operand_larger_than_4_gib(client, oracle):
    """Row offsets beyond 2^32 bytes: A is 270 000 x 8192 bf16 (4.4 GB), all zero except four planted rows."""
    m, n, k = 270_000, 512, 8192
    ha = client.empty(2 * m * k)
    client._s.check(client.lib.mi355_memset(client.ctx, None, ha.device_ptr(), 0, 2 * m * k))
    rows = [0, 131_071, 262_144 + 77, m - 1]                                   # the last two start beyond the 4 GiB mark
    assert rows[2] * k * 2 > 1 << 32
    a_rows = oracle.to_bf16(oracle.fill_uniform(len(rows) * k, 710, -1.0, 1.0)).reshape(len(rows), k)
    for r, bits in zip(rows, a_rows):
        client.write(ha.offset_start_by(2 * r * k).offset_end_by(2 * (m - r - 1) * k), bits)
    b = TensorHandle.uniform(client, (n, k), ElemType.BF16, SEED, 711, -1.0, 1.0)
    b_bits = oracle.to_bf16(oracle.fill_uniform(n * k, 711, -1.0, 1.0))
    a = TensorHandle.new_contiguous((m, k), ha, ElemType.BF16)
    c = TensorHandle.new_contiguous((m, n), client.empty(4 * m * n), ElemType.F32)
    d = N.GemmDesc(m=m, n=n, k=k, batch=1, lda=k, ldb=k, ldc=n, dtype_ab=N.DTYPE_BF16, dtype_c=N.DTYPE_F32, trans_b=1)
    assert ops.gemm_select(client, d) == N.GEMM_ALGO_LP_256M16                  # 2110 tiles of 256^2: the 16x16x32 form (same addressing)
    ops.matmul(client, a, TensorHandle.new(b.handle, (k, n), (1, k), ElemType.BF16), c)
    for r, bits in zip(rows, a_rows):
        got = client.read_one(c.handle.offset_start_by(4 * r * n).offset_end_by(4 * (m - r - 1) * n)).view(np.float32)
        _rows_check(oracle, bits, b_bits, got[None, :], np.array([0]), k, n, ElemType.BF16)
    for r in (1, 131_072, 262_144, m - 2):                                     # neighbours of the planted rows stay exactly zero
        got = client.read_one(c.handle.offset_start_by(4 * r * n).offset_end_by(4 * (m - r - 1) * n)).view(np.float32)
        assert not got.any()
    del a, c, ha
    client.memory_cleanup()


def _window(client, handle, start, count):
    """`count` bytes of a device buffer from byte `start` (a handle window, server/handle.rs:85-121)."""
    total = handle.size
    return client.read_one(handle.offset_start_by(start).offset_end_by(total - start - count))


def test_strided_copies_beyond_2_pow_32_elements(client):
    """copy_into at sizes whose linear indices do not fit 32 bits (the kernels' 64-bit index paths): rows, tiles, packed
    gather.  Data = counter-RNG bytes generated on the device; checked on sampled windows read back from both sides."""
    # rows: [8200, 2^20] bytes out of rows pitched by 4096 more -> 8.6e9 one-byte elements
    rows, cols, pitch = 8200, 1 << 20, (1 << 20) + 4096
    src = client.empty(rows * pitch)
    client._s.check(client.lib.mi355_fill_uniform(client.ctx, None, src.device_ptr(), N.DTYPE_U8, rows * pitch, SEED, 900, 0.0, 256.0))
    tin = TensorHandle.new(src, (rows, cols), (pitch, 1), ElemType.U8)
    out = ops.into_contiguous(client, tin)
    assert ops.copy_plan(client, tin, out) == (N.COPY_PATH_ROWS, 16)
    for r in (0, 1, 2047, 4095, 4096, 4097, 8199):          # rows either side of linear index 2^32 = row 4096
        a = _window(client, src, r * pitch, cols)
        b = _window(client, out.handle, r * cols, cols)
        assert a.any() and np.array_equal(a, b), r
    del out
    # the same with odd rows (2^20 - 1 bytes, pitch 2^20 + 4095): one-byte accesses, more than 2^32 of them
    todd = TensorHandle.new(src, (4400, cols - 1), (pitch - 1, 1), ElemType.U8)
    out = ops.into_contiguous(client, todd)
    assert ops.copy_plan(client, todd, out) == (N.COPY_PATH_ROWS, 1)
    for r in (0, 4095, 4096, 4097, 4399):
        a = _window(client, src, r * (pitch - 1), cols - 1)
        b = _window(client, out.handle, r * (cols - 1), cols - 1)
        assert np.array_equal(a, b), r
    del out, todd
    # packed gather: every second byte of the first 2 x (2^32 + 8192) bytes of the same buffer
    n = (1 << 32) + 8192
    tg = TensorHandle.new(src, (n,), (2,), ElemType.U8)
    g = ops.into_contiguous(client, tg)
    assert ops.copy_plan(client, tg, g) == (N.COPY_PATH_GENERIC, 4)
    for start in (0, (1 << 31) - 4096, (1 << 32) - 4096, (1 << 32) + 4096):
        a = _window(client, src, 2 * start, 8192)[::2]
        b = _window(client, g.handle, start, 4096)
        assert np.array_equal(a, b), start
    del g, tg, tin, src
    client.memory_cleanup()
    # tiles: [65536 + 256, 65536] one-byte transpose (4.3e9 elements), then back: the round trip is the identity
    p, q = 65536 + 256, 65536
    x = client.empty(p * q)
    client._s.check(client.lib.mi355_fill_uniform(client.ctx, None, x.device_ptr(), N.DTYPE_U8, p * q, SEED, 901, 0.0, 256.0))
    tx = TensorHandle.new_contiguous((q, p), x, ElemType.U8)               # stored [q][p]
    ty = ops.into_contiguous(client, tx.permute([1, 0]))                    # [p][q]
    assert ops.copy_plan(client, tx.permute([1, 0]), ty) == (N.COPY_PATH_TRANSPOSE, 16)
    for q0, p0 in ((0, 0), (65536 - 256, 65536), (40000, 12345), (65000, 65700)):
        blk = _window(client, x, q0 * p, 256 * p).reshape(256, p)           # input rows q0 .. q0 + 255
        for pp in (p0, p0 + 1, p0 + 77):
            got = _window(client, ty.handle, pp * q + q0, 256)              # output row pp, columns q0 .. q0 + 255
            assert np.array_equal(got, blk[:, pp]), (q0, pp)
    tz = ops.into_contiguous(client, ty.permute([1, 0]))
    for start in (0, (1 << 32) - 65536, p * q - 65536, 3 * (1 << 30) + 12345):
        assert np.array_equal(_window(client, x, start, 65536), _window(client, tz.handle, start, 65536)), start
    del tx, ty, tz, x
    client.memory_cleanup()


@pytest.mark.parametrize("m,n,k", [(8192, 8192, 64), (8192, 8192, 128), (1, 8192, 8192), (16, 8192, 8192), (64, 8192, 8192), (8192, 64, 8192)])
def test_skinny_and_output_bound_shapes_as_benched(client, oracle, m, n, k):
    """The bench's `gemm_bf16_shapes` entries that have kernels of their own, in the form bench.py times them (bf16 -> bf16,
    AUTO): the output-bound 8192 x 8192 x 64 / 128 take the 128x128 kernel in its single-stage form (four workgroups per CU),
    the GEMV the dot2 row-streaming kernel, the 16- and 64-row (or -column) shapes the no-split-K streaming kernel.  Sampled rows
    (all rows when there are at most 64) against the f64 oracle; for the short-K shapes every 128-row block position of a wave
    and the last tile are in the sample."""
    import ctypes as C
    a = TensorHandle.uniform(client, (m, k), ElemType.BF16, SEED, 700, -1.0, 1.0)
    b = TensorHandle.uniform(client, (n, k), ElemType.BF16, SEED, 701, -1.0, 1.0)
    c = TensorHandle.new_contiguous((m, n), client.empty(m * n * 2), ElemType.BF16)
    client._s.check(client.lib.mi355_memset(client.ctx, None, C.c_void_p(c.device_ptr()), 0xEE, m * n * 2))
    d = _bench_desc(m, n, k, N.DTYPE_BF16, N.DTYPE_BF16)
    want = N.GEMM_ALGO_SKINNY if m == 1 else N.GEMM_ALGO_STREAM64 if min(m, n) <= 64 else N.GEMM_ALGO_LP_128
    assert ops.gemm_select(client, d) == want
    client._s.check(client.lib.mi355_gemm(client.ctx, None, C.byref(d), C.c_void_p(a.device_ptr()), C.c_void_p(b.device_ptr()),
                                          C.c_void_p(c.device_ptr())))
    got = c.to_numpy(client).reshape(m, n)
    rows = np.arange(m) if m <= 64 else np.array([0, 31, 32, 63, 64, 127, 128, 4095, 4096 + 65, 8064, 8191])
    a_bits = oracle.to_bf16(oracle.fill_uniform(m * k, 700, -1.0, 1.0))
    b_bits = oracle.to_bf16(oracle.fill_uniform(n * k, 701, -1.0, 1.0))
    _bf16_rows_check(oracle, a_bits, b_bits, got[rows], rows, k, n)
    if want == N.GEMM_ALGO_LP_128:   # the one-tile-per-workgroup 256x256 kernel computes the same bits (same K order within a tile)
        c2 = TensorHandle.new_contiguous((m, n), client.empty(m * n * 2), ElemType.BF16)
        d.algo = N.GEMM_ALGO_LP_256W4
        client._s.check(client.lib.mi355_gemm(client.ctx, None, C.byref(d), C.c_void_p(a.device_ptr()), C.c_void_p(b.device_ptr()),
                                              C.c_void_p(c2.device_ptr())))
        assert np.array_equal(c2.to_numpy(client), c.to_numpy(client))


@pytest.mark.parametrize("m,n,k,want", [(16, 28672, 8192, "STREAM64"), (64, 28672, 8192, "LP_128"), (128, 14336, 4096, "LP_128")])
def test_decode_like_products_at_size(client, oracle, m, n, k, want):
    """tokens x out_features x in_features as a decoder serves them (bf16 -> bf16, AUTO): 16 tokens take the streaming kernel
    in its two-workgroups-per-CU form (896 workgroups), 64 and 128 tokens the 128x128 kernel WITHOUT split-K since round 2
    (224 / 112 tiles for 128 / 64 K-tiles: the launcher's rule, gemm_lp128.hip).  All rows, three blocks of 256 columns (first,
    one across a tile boundary in the middle, last) against the f64 oracle."""
    import ctypes as C
    a = TensorHandle.uniform(client, (m, k), ElemType.BF16, SEED, 710, -1.0, 1.0)
    b = TensorHandle.uniform(client, (n, k), ElemType.BF16, SEED, 711, -1.0, 1.0)
    c = TensorHandle.new_contiguous((m, n), client.empty(m * n * 2), ElemType.BF16)
    client._s.check(client.lib.mi355_memset(client.ctx, None, C.c_void_p(c.device_ptr()), 0xEE, m * n * 2))
    d = _bench_desc(m, n, k, N.DTYPE_BF16, N.DTYPE_BF16)
    assert ops.gemm_select(client, d) == getattr(N, "GEMM_ALGO_" + want)
    client._s.check(client.lib.mi355_gemm(client.ctx, None, C.byref(d), C.c_void_p(a.device_ptr()), C.c_void_p(b.device_ptr()),
                                          C.c_void_p(c.device_ptr())))
    got = c.to_numpy(client).reshape(m, n)
    a_bits = oracle.to_bf16(oracle.fill_uniform(m * k, 710, -1.0, 1.0))
    b_bits = oracle.to_bf16(oracle.fill_uniform(n * k, 711, -1.0, 1.0)).reshape(n, k)
    rows = np.arange(m)
    for c0 in (0, n // 2 - 128 - 64, n - 256):
        _bf16_rows_check(oracle, a_bits, b_bits[c0:c0 + 256].reshape(-1), got[:, c0:c0 + 256], rows, k, 256)
    assert not np.any(got == 0xEEEE)      # every element was written (0xEEEE is the fill pattern, -3.7e28 as a bf16)
