"""Pins the CPU oracle (oracle/oracle.c) against every known-answer vector the reference's own
tests hold for this path plus independent numpy restatements.  CPU only."""
import json
from pathlib import Path

import numpy as np
import pytest

GOLD = json.loads((Path(__file__).parent / "golden" / "reference_goldens.json").read_text())


def f16_bits(x, oracle):
    return oracle.to_f16(np.asarray(x, dtype=np.float32))


# ---- cmma fragment known answers ------------------------------------------------------------------
def test_cmma_simple_1_f16_nt(oracle):
    # runtime_tests/cmma.rs:500-501 inputs, :552-576 expectation; Out = Lhs @ Rhs.T (:23)
    lhs = f16_bits(np.arange(256), oracle)
    rhs = f16_bits(np.arange(256) % 8, oracle)
    out = oracle.gemm(lhs, rhs, 16, 16, 16, dtype_ab=oracle.DT_F16, trans_b=True)
    assert out.tolist() == GOLD["cmma_simple_1_f16_16x16x16_nt"]
    # closed form quoted in SURVEY.md: row r = 504 + 896 r
    assert out.reshape(16, 16)[:, 0].tolist() == [504.0 + 896.0 * r for r in range(16)]


def test_cmma_simple_tf32_nn(oracle):
    # cmma.rs:848-849 inputs; A 16x8 row-major, B 8x16 row-major (stride 16) (:219-237)
    lhs = np.arange(128, dtype=np.float32)
    rhs = (np.arange(128) % 8).astype(np.float32)
    out = oracle.gemm(lhs, rhs, 16, 16, 8, trans_b=False)
    assert out.tolist() == GOLD["cmma_simple_tf32_16x16x8_nn"]


def test_cmma_strided_lhs(oracle):
    # cmma.rs:946-956: (m, n, k) = (16, 16, 32), tiles 16^3, left K-half filled, lda = 32, ldb = 16
    m, n, k, tk = 16, 16, 32, 16
    i = np.arange(m * k)
    lhs = np.where((i % k) < tk, i - (i // k) * tk, 0).astype(np.float32)
    rhs = (np.arange(n * k) % 8).astype(np.float32)
    out = oracle.gemm(f16_bits(lhs, oracle), f16_bits(rhs, oracle), 16, 16, 16, dtype_ab=oracle.DT_F16,
                      lda=k, ldb=n, trans_b=True)
    assert out.tolist() == GOLD["cmma_strided_f16_16x16x16_nt"]


@pytest.mark.parametrize("m,n,k", [(16, 16, 16), (32, 32, 16), (32, 32, 32), (64, 64, 32), (16, 16, 64)])
def test_cmma_cube_expected(oracle, m, n, k):
    # test_simple_cube_expected (cmma.rs:695-722) restated in numpy with the same f32 loop order
    lhs = np.arange(m * k, dtype=np.float32)
    rhs = (np.arange(k * n) % 8).astype(np.float32)
    lhs16, rhs16 = f16_bits(lhs, oracle), f16_bits(rhs, oracle)
    lhs_f = lhs16.view(np.float16).astype(np.float32).reshape(m, k)
    rhs_f = rhs16.view(np.float16).astype(np.float32).reshape(n, k)
    expected = np.zeros((m, n), dtype=np.float32)
    for kk in range(k):
        expected += lhs_f[:, kk:kk + 1] * rhs_f[None, :, kk]
    out = oracle.gemm(lhs16, rhs16, m, n, k, dtype_ab=oracle.DT_F16, trans_b=True)
    assert np.array_equal(out.reshape(m, n), expected)


@pytest.mark.parametrize("m,n,k", [(16, 16, 16), (32, 32, 8), (16, 8, 16)])
def test_cmma_manual_row_major(oracle, m, n, k):
    # test_cmma_manual (cmma.rs:1127-1177): lhs[i,j] = 2i + j, rhs[i,j] = 3i + j, integer dot product
    lhs = np.array([[2 * i + j for j in range(k)] for i in range(m)], dtype=np.float32)
    rhs = np.array([[3 * i + j for j in range(n)] for i in range(k)], dtype=np.float32)
    expected = (lhs.astype(np.int64) @ rhs.astype(np.int64)).astype(np.float32)
    out = oracle.gemm(lhs, rhs, m, n, k, trans_b=False).reshape(m, n)
    assert np.array_equal(out, expected)
    out16 = oracle.gemm(oracle.to_bf16(lhs), oracle.to_bf16(rhs), m, n, k, dtype_ab=oracle.DT_BF16).reshape(m, n)
    assert np.allclose(out16, expected, rtol=0.03)  # the reference's own 3 % tolerance (:1183-1191)


def test_cmma_cast(oracle):
    # test_cmma_cast_f16 / _bf16 (cmma.rs:766-832): f32 0..255 -> 16-bit, exact
    x = np.arange(256, dtype=np.float32)
    assert np.array_equal(oracle.to_f16(x).view(np.float16), x.astype(np.float16))
    assert np.array_equal(oracle.from_bf16(oracle.to_bf16(x)), x)


# ---- 16-bit conversions against independent implementations ------------------------------------------
def test_f16_conversion_matches_numpy(oracle):
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(20000).astype(np.float32) * 10.0 ** rng.integers(-9, 6, 20000),
                        np.array([0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e-8, 5.96e-8, 2.98e-8, 2.9802322e-8,
                                  6.1e-5, np.inf, -np.inf], dtype=np.float32)]).astype(np.float32)
    with np.errstate(over="ignore"):
        assert np.array_equal(oracle.to_f16(x), x.astype(np.float16).view(np.uint16))
    h = np.arange(65536, dtype=np.uint16)
    ref = h.view(np.float16).astype(np.float32)
    got = oracle.from_f16(h)
    assert np.array_equal(np.isnan(ref), np.isnan(got))
    assert np.array_equal(ref[~np.isnan(ref)], got[~np.isnan(ref)])


def test_bf16_conversion_matches_torch(oracle):
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(50000) * 10.0 ** rng.integers(-20, 20, 50000)).astype(np.float32)
    ref = torch.from_numpy(x).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    assert np.array_equal(oracle.to_bf16(x), ref)


# ---- reductions -----------------------------------------------------------------------------------------
# ---- OCP FP8 (crates/cubecl-common/src/float/fp8/fp8_e4m3.rs, fp8_e5m2.rs; float8 0.7.0 underneath) --------
def test_fp8_known_answers_of_the_reference_tests(oracle):
    E4, E5 = oracle.DT_F8E4M3, oracle.DT_F8E5M2
    f = lambda *v: np.array(v, dtype=np.float32)
    # fp8_e4m3.rs:302-316 max_is_the_largest_finite_value / min_is_max_negated; MAX = 0x7E, NAN = S.1111.111 (:36-64)
    assert oracle.from_fp8(oracle.to_fp8(f(448.0), E4), E4)[0] == 448.0
    assert oracle.from_fp8(oracle.to_fp8(f(np.finfo(np.float32).max), E4), E4)[0] == 448.0
    assert oracle.to_fp8(f(448.0), E4)[0] == 0x7E and np.isnan(oracle.from_fp8(np.array([0x7F], np.uint8), E4)[0])
    assert oracle.from_fp8(np.array([0xFE], np.uint8), E4)[0] == -448.0
    assert oracle.to_fp8(f(np.inf, -np.inf), E4).tolist() == [0x7E, 0xFE]          # "infinities included, saturate"
    # fp8_e5m2.rs:307-335: MAX = 0x7B = 57344, the step past MAX is infinity, six NaN encodings (:115)
    assert oracle.from_fp8(oracle.to_fp8(f(57344.0), E5), E5)[0] == 57344.0
    assert oracle.from_fp8(oracle.to_fp8(f(np.finfo(np.float32).max), E5), E5)[0] == 57344.0
    assert oracle.to_fp8(f(57344.0), E5)[0] == 0x7B and np.isinf(oracle.from_fp8(np.array([0x7C], np.uint8), E5)[0])
    assert oracle.from_fp8(np.array([0xFB], np.uint8), E5)[0] == -57344.0
    nan5 = np.isnan(oracle.from_fp8(np.arange(256, dtype=np.uint8), E5))
    assert np.flatnonzero(nan5).tolist() == [0x7D, 0x7E, 0x7F, 0xFD, 0xFE, 0xFF]
    nan4 = np.isnan(oracle.from_fp8(np.arange(256, dtype=np.uint8), E4))
    assert np.flatnonzero(nan4).tolist() == [0x7F, 0xFF]
    # NaN stays NaN, zero keeps its sign, ONE = 0x38 / 0x3C
    for dt, one in ((E4, 0x38), (E5, 0x3C)):
        assert np.isnan(oracle.from_fp8(oracle.to_fp8(f(np.nan), dt), dt)[0])
        assert oracle.to_fp8(f(0.0, -0.0, 1.0, -1.0), dt).tolist() == [0x00, 0x80, one, 0x80 | one]


@pytest.mark.parametrize("name", ["e4m3", "e5m2"])
def test_fp8_conversion_matches_torch_and_round_trips(oracle, name):
    torch = pytest.importorskip("torch")
    dt, tdt, mx = {"e4m3": (oracle.DT_F8E4M3, torch.float8_e4m3fn, 448.0),
                   "e5m2": (oracle.DT_F8E5M2, torch.float8_e5m2, 57344.0)}[name]
    codes = np.arange(256, dtype=np.uint8)
    dec = oracle.from_fp8(codes, dt)
    # decode: every encoding equals torch's independent table (NaNs compared as NaNs)
    tdec = torch.arange(256, dtype=torch.uint8).view(tdt).float().numpy()
    assert np.array_equal(np.isnan(dec), np.isnan(tdec)) and np.array_equal(dec[~np.isnan(dec)], tdec[~np.isnan(tdec)])
    # every finite encoding round-trips; decoded values are strictly increasing over the positive codes
    fin = np.isfinite(dec)
    assert np.array_equal(oracle.to_fp8(dec[fin], dt), codes[fin])
    pos = dec[:128][np.isfinite(dec[:128])]
    assert np.all(np.diff(pos) > 0)
    # encode: round-to-nearest-even agrees with torch on random in-range values, on every midpoint between two
    # neighbouring encodings (the ties) and just beside them; torch does not saturate, so only |x| <= MAX
    rng = np.random.default_rng(7)
    mids = ((pos[:-1].astype(np.float64) + pos[1:].astype(np.float64)) / 2).astype(np.float32)
    x = np.concatenate([rng.uniform(-mx, mx, 100000), rng.normal(0, 1, 100000), rng.uniform(-2.0 ** -5, 2.0 ** -5, 100000),
                        mids, -mids, np.nextafter(mids, np.float32(0)), np.nextafter(mids, np.float32(1e9))]).astype(np.float32)
    x = x[np.abs(x) <= mx]
    ref = torch.from_numpy(x).to(tdt).view(torch.uint8).numpy()
    assert np.array_equal(oracle.to_fp8(x, dt), ref)
    # above MAX everything saturates (the reference's contract; torch's e4m3fn would give NaN there)
    big = np.array([mx * 1.0001, mx * 1.5, 1e30, -1e30], dtype=np.float32)
    assert oracle.to_fp8(big, dt).tolist() == [oracle.to_fp8(np.float32([mx]), dt)[0]] * 3 + [0x80 | oracle.to_fp8(np.float32([mx]), dt)[0]]


def test_fp8_gemm_is_exact_in_f32_for_small_k(oracle):
    # products of two 4-bit significands are exact in f32 and so are short sums of them: the f32 loop of
    # test_simple_cube_expected (cmma.rs:695-722) and the f64 oracle agree bit for bit
    rng = np.random.default_rng(3)
    m, n, k = 8, 12, 32
    a = oracle.to_fp8(rng.uniform(-1, 1, m * k).astype(np.float32))
    b = oracle.to_fp8(rng.uniform(-1, 1, n * k).astype(np.float32))
    c32 = oracle.gemm(a, b, m, n, k, dtype_ab=oracle.DT_F8E4M3, trans_b=True)
    c64 = oracle.gemm(a, b, m, n, k, dtype_ab=oracle.DT_F8E4M3, trans_b=True, acc_f64=True)
    ref = oracle.from_fp8(a).reshape(m, k).astype(np.float64) @ oracle.from_fp8(b).reshape(n, k).astype(np.float64).T
    assert np.array_equal(c64.reshape(m, n), ref.astype(np.float32))
    assert np.allclose(c32, c64, rtol=0, atol=1e-5)


# ---- block-scaled MMA (runtime_tests/cmma.rs:1476-1704) -------------------------------------------------------------
def _scaled_case(oracle, m, n, k, factor, fp4=False):
    """The reference test's data (cmma.rs:1517-1531 / :1624-1641), generated exactly as written there."""
    i, j = np.meshgrid(np.arange(m), np.arange(k), indexing="ij")
    jn, ik = np.meshgrid(np.arange(n), np.arange(k), indexing="ij")
    if fp4:
        lhs = oracle.unpack_e2m1x2(np.arange(16, dtype=np.uint8) * 0x11)[::2][((i + j) % 15) + 1]     # e2m1::from_bits(((i+j)%15)+1)
        rhs = oracle.unpack_e2m1x2(np.arange(16, dtype=np.uint8) * 0x11)[::2][((ik + jn) % 15) + 1]
    else:
        lhs = (i * 2 + j).astype(np.float32)
        rhs = (ik * 3 + jn).astype(np.float32)
    si, sj = np.meshgrid(np.arange(m), np.arange(factor), indexing="ij")
    lhs_scales = (si * 2 + sj + 120).astype(np.uint8)
    sjn, sif = np.meshgrid(np.arange(n), np.arange(factor), indexing="ij")
    rhs_scales = (sif * 3 + sjn + 120).astype(np.uint8)
    return lhs.astype(np.float32), lhs_scales, rhs.astype(np.float32), rhs_scales


def _scaled_expected(lhs, ls, rhs, rs, m, n, k, factor):
    """cmma.rs:1572-1591 verbatim in numpy: f32, left to right."""
    out = np.zeros((m, n), dtype=np.float32)
    for l in range(k):
        blk = l // (k // factor)
        p = (lhs[:, l:l + 1] * ls[:, blk:blk + 1]).astype(np.float32)
        p = (p * rhs[None, :, l]).astype(np.float32)
        p = (p * rs[None, :, blk]).astype(np.float32)
        out = (out + p).astype(np.float32)
    return out


@pytest.mark.parametrize("dt", ["e4m3", "e5m2"])
def test_scaled_mma_fp8_reference_case(oracle, dt):
    m, n, k, factor = 16, 8, 32, 1                                        # cmma.rs:1913-1916
    dtype = {"e4m3": oracle.DT_F8E4M3, "e5m2": oracle.DT_F8E5M2}[dt]
    lhs, lsc, rhs, rsc = _scaled_case(oracle, m, n, k, factor)
    a, b = oracle.to_fp8(lhs, dtype), oracle.to_fp8(rhs, dtype)
    got = oracle.gemm_scaled(a, lsc, b, rsc, m, n, k, dtype_ab=dtype, block=k // factor).reshape(m, n)
    # bit-exact against the loop evaluated on the values the fp8 buffers really hold ...
    want = _scaled_expected(oracle.from_fp8(a, dtype), oracle.from_ue8m0(lsc), oracle.from_fp8(b, dtype), oracle.from_ue8m0(rsc),
                            m, n, k, factor)
    assert np.array_equal(got, want)
    # ... and, like the reference (assert_equals_approx 0.03, :1593), close to the loop on the unrounded integers
    # (the reference's own budget for the A::from(i*2+j) rounding; e5m2 keeps 2 mantissa bits, so only e4m3 fits it)
    ideal = _scaled_expected(lhs, oracle.from_ue8m0(lsc), rhs, oracle.from_ue8m0(rsc), m, n, k, factor)
    if dt == "e4m3":
        assert np.all(np.abs(got - ideal) <= 0.03 * np.abs(ideal) + 1e-6)


def test_scaled_mma_fp4_reference_case(oracle):
    m, n, k, factor = 16, 8, 64, 2                                        # cmma.rs:1936
    lhs, lsc, rhs, rsc = _scaled_case(oracle, m, n, k, factor, fp4=True)
    a, b = oracle.pack_e2m1x2(lhs), oracle.pack_e2m1x2(rhs)
    assert np.array_equal(oracle.unpack_e2m1x2(a), lhs.reshape(-1))        # every value of the test is an e2m1 value
    got = oracle.gemm_scaled(a, lsc, b, rsc, m, n, k, dtype_ab=oracle.DT_F4E2M1X2, block=k // factor).reshape(m, n)
    want = _scaled_expected(lhs, oracle.from_ue8m0(lsc), rhs, oracle.from_ue8m0(rsc), m, n, k, factor)
    assert np.array_equal(got, want)


def test_e2m1_and_ue8m0_tables(oracle):
    # OCP MX value table of e2m1 (e2m1::MAX = 0x7 = 6.0, MIN = 0xf, fp4.rs:31-35); low nibble first (fp4.rs:204-224)
    vals = oracle.unpack_e2m1x2(np.arange(16, dtype=np.uint8) * 0x11)[::2]
    assert vals.tolist() == [0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0, -0.0, -0.5, -1.0, -1.5, -2.0, -3.0, -4.0, -6.0]
    assert oracle.pack_e2m1x2(np.float32([1.0, -6.0, 0.5])).tolist() == [0xF2, 0x01]
    assert oracle.unpack_e2m1x2(np.uint8([0xF2])).tolist() == [1.0, -6.0]
    # round to nearest, ties to the even code, saturating
    x = np.float32([0.24, 0.25, 0.26, 0.75, 1.25, 1.75, 2.5, 3.5, 5.0, 5.1, 7.0, 1e9, -0.75])
    assert oracle.unpack_e2m1x2(oracle.pack_e2m1x2(x), x.size).tolist() == [0.0, 0.0, 0.5, 1.0, 1.0, 2.0, 2.0, 4.0, 4.0, 6.0, 6.0, 6.0, -1.0]
    s = oracle.from_ue8m0(np.uint8([0, 1, 126, 127, 128, 254, 255]))
    assert s[:6].tolist() == [2.0 ** -127, 2.0 ** -126, 0.5, 1.0, 2.0, 2.0 ** 127] and np.isnan(s[6])


def test_sum_things_input(oracle):
    # examples/sum_things/src/lib.rs:180: [-1, 10, 1, 5] -> 15
    x = np.array([-1.0, 10.0, 1.0, 5.0], dtype=np.float32)
    assert oracle.sum_sequential(x) == 15.0
    assert oracle.sum_f64(x) == 15.0


def test_book_reduce_matrix(oracle):
    # cubecl-book/src/getting-started/src/bin/v1-cpu.rs:3-6: arange 3x3 -> [3, 12, 21]
    x = np.arange(9, dtype=np.float32).reshape(3, 3)
    assert oracle.reduce_last_axis_sum(x).tolist() == [3.0, 12.0, 21.0]


@pytest.mark.parametrize("vec", [1, 2, 4])
def test_plane_sum_reference_vectors(oracle, vec):
    # test_plane_sum (runtime_tests/plane.rs:154-190): plane_size 32, input 0..32*vec,
    # expected[v] = sum_k input[v + k*vec]; tolerance 1e-5 relative (binary.rs:15-53)
    plane = 32
    x = np.arange(plane * vec, dtype=np.float32).reshape(plane, vec)
    for v in range(vec):
        lanes = oracle.plane_reduce(x[:, v], 0)
        expected = x[:, v].sum(dtype=np.float64)
        assert np.allclose(lanes, expected, rtol=1e-5)
        assert np.all(lanes == lanes[0])  # every lane holds the result


def test_plane_ops_butterfly(oracle):
    rng = np.random.default_rng(2)
    v = rng.standard_normal(64).astype(np.float32)
    assert np.allclose(oracle.plane_reduce(v, 0), v.sum(dtype=np.float64), rtol=1e-5)
    assert np.all(oracle.plane_reduce(v, 2) == v.max())
    assert np.all(oracle.plane_reduce(v, 3) == v.min())
    inc = oracle.plane_inclusive_sum(np.arange(32, dtype=np.float32))
    assert inc.tolist() == np.cumsum(np.arange(32)).astype(np.float32).tolist()  # plane.rs:192-230


@pytest.mark.parametrize("vec", [1, 2, 4])
def test_plane_prod_scans_reproduce_the_reference_tests_expectations(oracle, vec):
    """test_plane_inclusive_prod / test_plane_exclusive_prod (crates/cubecl-core/src/runtime_tests/plane.rs:317-407) restated:
    plane_size 32, input x % 3 -> 0.5 / 1.25 / 1.75, expected[k] = product of the inputs of the units to the left (inclusive:
    and its own; exclusive: unit 0 holds 1), per vector component; the reference compares with 1e-5 x max(|e|, 1) scaled to the
    dtype's epsilon (:852-880).  And the two doc examples of frontend/plane.rs:304, :329."""
    plane = 32
    x = np.array([(0.5, 1.25, 1.75)[i % 3] for i in range(plane * vec)], dtype=np.float32).reshape(plane, vec)
    for v in range(vec):
        inc_expected = x[:, v].astype(np.float32).copy()
        exc_expected = np.ones(plane, dtype=np.float32)
        for k in range(1, plane):                              # the reference test's own (sequential, f32) loops
            for k1 in range(k):
                inc_expected[k] = np.float32(inc_expected[k] * x[k1, v])
                exc_expected[k] = np.float32(exc_expected[k] * x[k1, v])
        inc = oracle.plane_scan(x[:, v], mul=True, exclusive=False)
        exc = oracle.plane_scan(x[:, v], mul=True, exclusive=True)
        assert np.all(np.abs(inc - inc_expected) <= 1e-5 * np.maximum(np.abs(inc_expected), 1.0))
        assert np.all(np.abs(exc - exc_expected) <= 1e-5 * np.maximum(np.abs(exc_expected), 1.0))
        assert exc[0] == 1.0 and np.array_equal(exc[1:], inc[:-1])            # plane_reduce_exclusive = inclusive shuffled up (shared/plane.rs:89-97)
    assert oracle.plane_scan(np.arange(1, 6), True, False).tolist() == [1, 2, 6, 24, 120]
    assert oracle.plane_scan(np.arange(1, 6), True, True).tolist() == [1, 1, 2, 6, 24]
    assert oracle.plane_scan(np.arange(32), False, False).tolist() == oracle.plane_inclusive_sum(np.arange(32, dtype=np.float32)).tolist()
    assert oracle.plane_scan(np.arange(1, 6), False, True).tolist() == [0, 1, 3, 6, 10]


def test_value_and_argmin_rules(oracle):
    """The rules of mi355_reduce / mi355_argreduce (include/mi355cube.h): max / min propagate NaN and order -0 below +0; argmin
    mirrors argmax (lowest index, -0 == +0, first NaN wins); test_plane_max / _min's vectors (runtime_tests/plane.rs:409-487:
    0..32 with element 16 set to 999 / -5)."""
    x = np.arange(32, dtype=np.float32)
    x[16] = 999.0
    assert oracle.reduce_value(x, "max") == 999.0 and oracle.plane_reduce(x, 2)[0] == 999.0
    x[16] = -5.0
    assert oracle.reduce_value(x, "min") == -5.0 and oracle.plane_reduce(x, 3)[0] == -5.0
    y = np.array([3.0, -0.0, 0.0, -7.0, 3.0, -7.0], dtype=np.float32)
    assert oracle.argmin(y) == (3, np.float32(-7.0)) and oracle.argmax(y)[0] == 0
    z = np.array([0.0, -0.0, 0.0], dtype=np.float32)
    assert oracle.argmin(z)[0] == 0 and oracle.argmax(z)[0] == 0                       # -0 == +0: the lowest index
    assert not np.signbit(oracle.reduce_value(z, "max")) and np.signbit(oracle.reduce_value(z, "min"))   # as values: -0 < +0
    w = np.array([1.0, np.nan, -5.0, np.nan], dtype=np.float32)
    assert oracle.argmin(w)[0] == 1 and oracle.argmax(w)[0] == 1                       # the first NaN wins both
    assert np.isnan(oracle.reduce_value(w, "max")) and np.isnan(oracle.reduce_value(w, "min"))
    e = np.zeros(0, dtype=np.float32)
    assert oracle.reduce_value(e, "max") == -np.inf and oracle.reduce_value(e, "min") == np.inf
    assert oracle.argmin(e) == (0, np.float32(np.inf)) and oracle.reduce_value(e, "prod") == 1.0 and oracle.reduce_value(e, "mean") == 0.0
    m = np.arange(24, dtype=np.float32).reshape(2, 3, 4)
    assert oracle.reduce_axis_value(m, 1, "mean").tolist() == m.astype(np.float64).mean(axis=1).tolist()
    assert oracle.reduce_axis_value(m, 2, "prod")[0].tolist() == [0.0, 840.0, 7920.0]
    assert oracle.reduce_axis_argmin(m, 0).tolist() == np.zeros((3, 4)).tolist()
    assert oracle.reduce_axis_value(-m, 1, "max").tolist() == (-m).max(axis=1).tolist()


def test_argmax_rules(oracle):
    x = np.array([1.0, 7.0, 3.0, 7.0, -2.0], dtype=np.float32)
    assert oracle.argmax(x) == (1, np.float32(7.0))            # lowest index among equal maxima
    x = np.array([-0.0, 0.0, -1.0], dtype=np.float32)
    assert oracle.argmax(x)[0] == 0                             # -0.0 == +0.0
    x = np.array([1.0, np.nan, 5.0, np.nan], dtype=np.float32)
    assert oracle.argmax(x)[0] == 1                             # NaN ranks highest, first NaN wins
    x = np.array([-np.inf, -np.inf], dtype=np.float32)
    assert oracle.argmax(x)[0] == 0
    rng = np.random.default_rng(3)
    y = rng.standard_normal(100003).astype(np.float32)
    assert oracle.argmax(y)[0] == int(np.argmax(y))


def test_all_reduce_closed_form():
    # runtime_tests/all_reduce.rs:24-59: device i holds handles filled with (i + j); after
    # all_reduce(Sum) every device reads sum(ids) + j * device_count
    for ndev in (2, 4, 8):
        for j in range(8):
            contributions = [np.full(100, i + j, dtype=np.float32) for i in range(ndev)]
            total = np.sum(contributions, axis=0)
            assert np.all(total == sum(range(ndev)) + j * ndev)


# ---- synthetic data generator pin ------------------------------------------------------------------------
def test_rng_pinned(oracle):
    x = oracle.fill_uniform(8, 1, 0.0, 1.0)
    pinned = json.loads((Path(__file__).parent / "golden" / "oracle_pins.json").read_text())
    assert x.view(np.uint32).tolist() == pinned["fill_uniform_t1_0_1_first8_bits"]
    y = oracle.fill_uniform(1 << 16, 7, -1.0, 1.0)
    assert y.min() >= -1.0 and y.max() < 1.0 and abs(float(y.mean())) < 0.02
    z = oracle.fill_uniform(1 << 16, 8, -1.0, 1.0)
    assert not np.array_equal(y, z)  # tensor id decorrelates streams


def test_sequential_vs_f64_sum_gap(oracle):
    # a literal sequential-f32 sum drifts; the f64 oracle is the one of record
    x = oracle.fill_uniform(1 << 20, 3, 0.0, 1.0)
    exact = oracle.sum_f64(x)
    assert abs(exact - float(np.sum(x, dtype=np.float64))) < 1e-6 * exact
    assert abs(oracle.sum_sequential(x) - exact) / exact < 1e-3


# ---- strided copies: oracle/layout.py against the reference tests' known answers and CPU formulas ----------------------
def test_copy_into_rank_mismatch_known_answer():
    # tests/tensor/into_contiguous.rs:139-185: data 1..8, shape [1, 2, 4, 1] strides [8, 1, 2, 2] -> [1, 2, 4] contiguous;
    # the test's formula gives src(q) = (q / 4) % 2 + 2 * (q % 4)
    from oracle import layout as L
    data = np.arange(1, 9, dtype=np.float32)
    out = L.copy_into(data, [1, 2, 4, 1], [8, 1, 2, 2], np.zeros(8, np.float32), [1, 2, 4], [8, 4, 1])
    assert out.tolist() == [1, 3, 5, 7, 2, 4, 6, 8]


def test_copy_into_permuted_sweep_matches_numpy_transpose():
    # :257-283: every permutation of the reference's shapes; payload (i % 251) + 1 (:204-206).  numpy's transpose is an
    # independent statement of "element q of the permuted view".
    import itertools
    from oracle import layout as L
    shapes = [[2, 3], [3, 2], [4, 6], [6, 4], [2, 2, 3], [3, 2, 2], [2, 3, 4], [4, 3, 2], [8, 3, 4], [3, 8, 5], [16, 5], [5, 16],
              [8, 8], [16, 16], [32, 2, 3], [2, 3, 4, 5]]
    for shape in shapes:
        n = int(np.prod(shape))
        data = ((np.arange(n) % 251) + 1).astype(np.uint8)
        bstr = L.contiguous_strides(shape)
        for perm in itertools.permutations(range(len(shape))):
            got = L.into_contiguous(data, [shape[p] for p in perm], [bstr[p] for p in perm])
            assert np.array_equal(got, data.reshape(shape).transpose(perm).reshape(-1)), (shape, perm)


def test_copy_into_strided_destination_and_broadcast():
    from oracle import layout as L
    src = np.arange(12, dtype=np.int32)
    dst = np.full(20, -1, dtype=np.int32)
    L.copy_into(src, [3, 4], [4, 1], dst, [3, 4], [6, 1])                 # pitched rows: the padding stays untouched
    assert dst.tolist() == [0, 1, 2, 3, -1, -1, 4, 5, 6, 7, -1, -1, 8, 9, 10, 11, -1, -1, -1, -1]
    out = L.into_contiguous(np.array([7, 8, 9], dtype=np.int32), [2, 3], [0, 1])   # stride 0 broadcasts
    assert out.tolist() == [7, 8, 9, 7, 8, 9]


@pytest.mark.parametrize("shape,dim", [([1, 8, 16], 1), ([1, 8, 8], 1), ([4096, 256], 0), ([8192, 32], 0)])
def test_packed_repack_reference_cases(shape, dim):
    # :120-142: the four repack cases; expected = the test's own CPU packer applied along the innermost axis
    from oracle import layout as L
    n = int(np.prod(shape))
    unpacked = ((np.arange(n) % 15) + 1).astype(np.uint32)
    storage = L.pack_along(unpacked, shape, dim, 8, 4).astype(np.uint32)
    in_shape = list(shape)
    in_shape[dim] = -(-in_shape[dim] // 8)
    got = L.into_contiguous_packed(storage, L.contiguous_strides(in_shape), shape, len(shape) - 1 - dim, 8)
    assert got.any() and np.array_equal(got, L.pack_along(unpacked, shape, len(shape) - 1, 8, 4).astype(np.uint32))


def test_pack_along_known_words():
    # by hand: values 1..8 (4 bits each) along the only axis -> 0x87654321; packed along axis 0 of [8, 2]: column-wise
    from oracle import layout as L
    assert L.pack_along(np.arange(1, 9, dtype=np.uint32), [8], 0, 8, 4).tolist() == [0x87654321]
    v = np.arange(1, 17, dtype=np.uint32) & 15
    assert L.pack_along(v, [8, 2], 0, 8, 4).tolist() == [0xFDB97531, 0x0ECA8642]


def test_remaining_plane_intrinsics_reproduce_the_reference_tests_expectations(oracle):
    """runtime_tests/plane.rs:527-850 restated: the inputs those tests build and the `expected` vectors they compute, against
    the oracle's plane_op (all / any / elect / broadcast / ballot / shuffle / xor / up / down)."""
    plane = 32
    x = (np.arange(plane) % 5).astype(np.float32)
    x[4] = 10.0                                                      # test_plane_all / _any (:527-605), vectorization 1
    assert np.array_equal(oracle.plane_op((x < 5).astype(np.float32), oracle.PLANE_ALL, plane), np.zeros(plane, np.float32))
    assert np.array_equal(oracle.plane_op((x > 5).astype(np.float32), oracle.PLANE_ANY, plane), np.ones(plane, np.float32))
    y = (np.arange(plane) % 5).astype(np.float32)                    # ... and the untouched odd batches: all true / none true
    assert np.array_equal(oracle.plane_op((y < 5).astype(np.float32), oracle.PLANE_ALL, plane), np.ones(plane, np.float32))
    assert np.array_equal(oracle.plane_op((y > 5).astype(np.float32), oracle.PLANE_ANY, plane), np.zeros(plane, np.float32))
    # test_plane_ballot (:607-629): UNIT_POS < 8 on a 32-unit cube -> [0b1111_1111, 0, 0, 0]
    assert oracle.plane_op((np.arange(32) < 8).astype(np.float32), oracle.PLANE_BALLOT, 32).tolist() == [[0b11111111, 0, 0, 0]]
    # test_plane_elect (:631-659): exactly one unit of the plane adds 1 to output[20]
    assert oracle.plane_op(np.zeros(32, np.float32), oracle.PLANE_ELECT, 32).sum() == 1.0
    v = np.arange(plane, dtype=np.float32)
    assert oracle.plane_op(v, oracle.PLANE_BROADCAST, plane, 2)[0] == v[2]                      # :661-694: unit 0 receives lane 2
    assert oracle.plane_op(np.arange(64, dtype=np.float32), oracle.PLANE_SHUFFLE, 64, 0)[0] == 0.0   # :696-729
    assert np.array_equal(oracle.plane_op(v, oracle.PLANE_SHUFFLE_XOR, plane, 1), v[np.arange(plane) ^ 1])          # :731-770
    up = v.copy(); up[1:] = v[:-1]
    assert np.array_equal(oracle.plane_op(v, oracle.PLANE_SHUFFLE_UP, plane, 1), up)                                  # :772-810: lane 0 keeps its value
    w = np.arange(64, dtype=np.float32)
    down = w.copy(); down[:-1] = w[1:]
    assert np.array_equal(oracle.plane_op(w, oracle.PLANE_SHUFFLE_DOWN, 64, 1), down)                                 # :812-850: the last lane keeps its value
