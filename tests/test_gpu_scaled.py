"""Block-scaled (MX) matmul through the C ABI: mi355_gemm_scaled against the oracle's restatement of
test_cmma_scaled / test_cmma_scaled_fp4 (crates/cubecl-core/src/runtime_tests/cmma.rs:1476-1704).

Tolerances: the scalar kernel follows the reference loop literally -> bit-exact against the oracle's f32 loop.  The MFMA
kernel (v_mfma_scale_f32_32x32x64_f8f6f4) is compared with C_ref accumulated in f64 from the same bytes:
  * scales of equal magnitude inside a 64-wide step: |C - C_ref| <= 1e-5 * sum|a sa b sb| (BASELINE.json's f32 bound);
  * scales spread over 2^-9 .. 2^9 per operand (products over 2^+-18 inside one 64-term hardware dot product): 1e-4 *
    sum|a sa b sb| -- the matrix core aligns the 64 products of a step to the largest one and drops what falls below
    its internal width (measured: up to 2.1e-5 of sum|..|); the reference's own tolerance for this op is 3 %
    (assert_equals_approx 0.03, cmma.rs:1593);
  * 16-bit outputs: one ulp of the output format on top."""
import ctypes as C

import numpy as np
import pytest

from cubecl_amd import ElemType, ServerError, TensorHandle, ops
from cubecl_amd import _native as N

pytestmark = pytest.mark.gpu
REL = 1e-5          # equal-magnitude scales
REL_WIDE = 1e-4     # scales spread over 2^+-9 per operand (see the module docstring)
E4, E5, F4 = ElemType.F8E4M3, ElemType.F8E5M2, ElemType.F4E2M1X2


def _encode(oracle, x, dtype):
    """f32 array [rows, k] -> (device bytes [rows, k or k/2], the values those bytes hold)."""
    if dtype == F4:
        bits = oracle.pack_e2m1x2(x).reshape(x.shape[0], x.shape[1] // 2)
        return bits, oracle.unpack_e2m1x2(bits).reshape(x.shape)
    bits = oracle.to_fp8(x, int(dtype))
    return bits, oracle.from_fp8(bits, int(dtype))


def run_scaled(client, oracle, m, n, k, da, db, out, *, block=32, batch=1, bcast_b=False, algo=N.GEMM_ALGO_AUTO, scale_lo=118,
               scale_hi=137, seed=11, amp=None, exact=False, rel=None):
    rel = rel if rel is not None else (REL if scale_hi - scale_lo <= 1 else REL_WIDE)
    rng = np.random.default_rng(seed)
    amp = amp or (6.0 if da == F4 else 4.0)
    nb = k // block
    a_host = rng.uniform(-amp, amp, (batch, m, k)).astype(np.float32)
    b_host = rng.uniform(-amp, amp, (1 if bcast_b else batch, n, k)).astype(np.float32)
    sa = rng.integers(scale_lo, scale_hi, (batch, m, nb)).astype(np.uint8)
    sb = rng.integers(scale_lo, scale_hi, (1 if bcast_b else batch, n, nb)).astype(np.uint8)
    a_bits, a_val = zip(*[_encode(oracle, a_host[i], da) for i in range(a_host.shape[0])])
    b_bits, b_val = zip(*[_encode(oracle, b_host[i], db) for i in range(b_host.shape[0])])
    a_bits, b_bits = np.stack(a_bits), np.stack(b_bits)
    ta = TensorHandle.from_numpy(client, a_bits, da)
    tb = TensorHandle.from_numpy(client, b_bits, db)
    if bcast_b:
        tb = TensorHandle.new(tb.handle, (batch,) + b_bits.shape[1:], (0,) + tb.strides[1:], db)
    tsa = TensorHandle.from_numpy(client, sa, ElemType.UE8M0)
    tsb = TensorHandle.from_numpy(client, sb, ElemType.UE8M0)
    if bcast_b:
        tsb = TensorHandle.new(tsb.handle, (batch,) + sb.shape[1:], (0,) + tsb.strides[1:], ElemType.UE8M0)
    c_h = client.empty(batch * m * n * out.size())
    client._s.check(client.lib.mi355_memset(client.ctx, None, c_h.device_ptr(), 0xEE, c_h.size))
    tc = TensorHandle.new_contiguous((batch, m, n), c_h, out)
    ops.matmul_scaled(client, ta, tsa, tb, tsb, tc, block=block, algo=algo)
    raw = client.read_one(c_h)
    got_all = raw.view(np.float32 if out == ElemType.F32 else np.uint16).reshape(batch, m, n)
    for bi in range(batch):
        bj = 0 if bcast_b else bi
        if exact:      # the scalar kernel: same loop, same roundings
            want = oracle.gemm_scaled(a_bits[bi], sa[bi], b_bits[bj], sb[bj], m, n, k, dtype_ab=int(da), block=block,
                                      dtype_c=int(out)) if da == db else None
            if want is not None:
                assert np.array_equal(got_all[bi].reshape(-1), want)
                continue
        A = a_val[bi].astype(np.float64) * np.repeat(oracle.from_ue8m0(sa[bi]).astype(np.float64), block, axis=1)
        B = b_val[bj].astype(np.float64) * np.repeat(oracle.from_ue8m0(sb[bj]).astype(np.float64), block, axis=1)
        ref = A @ B.T
        bound = np.abs(A) @ np.abs(B).T
        if out == ElemType.F32:
            got = got_all[bi].astype(np.float64)
            err = np.abs(got - ref)
            assert np.all(err <= rel * bound + 1e-30), float((err / (bound + 1e-30)).max())
        else:
            got = (oracle.from_bf16(got_all[bi]) if out == ElemType.BF16 else oracle.from_f16(got_all[bi])).astype(np.float64)
            ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(ref), 1e-30))) - (7 if out == ElemType.BF16 else 10))
            assert np.all(np.abs(got - ref) <= ulp + rel * bound)


# ---- the reference's own cases ----------------------------------------------------------------------------------------
def _reference_case(oracle, m, n, k, factor, fp4):
    i, j = np.meshgrid(np.arange(m), np.arange(k), indexing="ij")
    jn, ik = np.meshgrid(np.arange(n), np.arange(k), indexing="ij")
    if fp4:
        table = oracle.unpack_e2m1x2(np.arange(16, dtype=np.uint8) * 0x11)[::2]
        lhs, rhs = table[((i + j) % 15) + 1], table[((ik + jn) % 15) + 1]          # cmma.rs:1624-1636
    else:
        lhs, rhs = (i * 2 + j).astype(np.float32), (ik * 3 + jn).astype(np.float32)  # cmma.rs:1517-1531
    si, sj = np.meshgrid(np.arange(m), np.arange(factor), indexing="ij")
    sjn, sif = np.meshgrid(np.arange(n), np.arange(factor), indexing="ij")
    return lhs.astype(np.float32), (si * 2 + sj + 120).astype(np.uint8), rhs.astype(np.float32), (sif * 3 + sjn + 120).astype(np.uint8)


@pytest.mark.parametrize("da,db", [(E5, E5), (E4, E4), (E5, E4), (E4, E5)])            # cmma.rs:1913-1916
def test_reference_scaled_fp8_case(client, oracle, da, db):
    m, n, k, factor = 16, 8, 32, 1
    lhs, lsc, rhs, rsc = _reference_case(oracle, m, n, k, factor, False)
    a, b = oracle.to_fp8(lhs, int(da)), oracle.to_fp8(rhs, int(db))
    c = TensorHandle.new_contiguous((m, n), client.empty(m * n * 4), ElemType.F32)
    ops.matmul_scaled(client, TensorHandle.from_numpy(client, a, da), TensorHandle.from_numpy(client, lsc, ElemType.UE8M0),
                      TensorHandle.from_numpy(client, b, db), TensorHandle.from_numpy(client, rsc, ElemType.UE8M0), c, block=k // factor)
    got = c.to_numpy(client)
    av, bv = oracle.from_fp8(a, int(da)).astype(np.float64), oracle.from_fp8(b, int(db)).astype(np.float64)
    ls, rs = oracle.from_ue8m0(lsc).astype(np.float64), oracle.from_ue8m0(rsc).astype(np.float64)
    want = ((av * ls) @ (bv * rs).T).astype(np.float32)      # exact in f64, and the f32 loop of these magnitudes is exact too
    assert np.array_equal(got, want)
    if da == E4 and db == E4:                                  # the reference's 3 % against the unrounded integers (:1593)
        ideal = (lhs.astype(np.float64) * ls) @ (rhs.astype(np.float64) * rs).T
        assert np.all(np.abs(got - ideal) <= 0.03 * np.abs(ideal) + 1e-9)


def test_reference_scaled_fp4_case(client, oracle):
    m, n, k, factor = 16, 8, 64, 2                                                       # cmma.rs:1936
    lhs, lsc, rhs, rsc = _reference_case(oracle, m, n, k, factor, True)
    a, b = oracle.pack_e2m1x2(lhs).reshape(m, k // 2), oracle.pack_e2m1x2(rhs).reshape(n, k // 2)
    c = TensorHandle.new_contiguous((m, n), client.empty(m * n * 4), ElemType.F32)
    ops.matmul_scaled(client, TensorHandle.from_numpy(client, a, F4), TensorHandle.from_numpy(client, lsc, ElemType.UE8M0),
                      TensorHandle.from_numpy(client, b, F4), TensorHandle.from_numpy(client, rsc, ElemType.UE8M0), c, block=k // factor)
    want = oracle.gemm_scaled(a, lsc, b, rsc, m, n, k, dtype_ab=int(F4), block=k // factor)
    assert np.array_equal(c.to_numpy(client).reshape(-1), want)


# ---- the scalar kernel: any block size, any shape -------------------------------------------------------------------------
@pytest.mark.parametrize("da,m,n,k,block", [(E4, 33, 17, 96, 32), (E5, 8, 40, 64, 16), (F4, 21, 19, 128, 64), (F4, 5, 3, 32, 32),
                                            (E4, 7, 9, 24, 1)])
def test_scalar_kernel_is_bit_exact(client, oracle, da, m, n, k, block):
    run_scaled(client, oracle, m, n, k, da, da, ElemType.F32, block=block, algo=N.GEMM_ALGO_GENERIC, exact=True)


# ---- the MFMA kernel ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("da,db", [(E4, E4), (E5, E5), (E4, E5), (E5, E4), (F4, F4)])
@pytest.mark.parametrize("out", [ElemType.F32, ElemType.BF16])
@pytest.mark.parametrize("m,n,k", [(256, 256, 256), (256, 512, 1024), (300, 504, 512), (16, 4096, 512)])
def test_mfma_scaled_parity(client, oracle, da, db, out, m, n, k):
    d = N.GemmScaledDesc(m=m, n=n, k=k, batch=1, lda=k, ldb=k, ldc=n, ld_sa=k // 32, ld_sb=k // 32, dtype_a=int(da), dtype_b=int(db),
                         dtype_c=int(out), block=32)
    assert ops.gemm_scaled_select(client, d) == N.GEMM_ALGO_LP_256W4
    run_scaled(client, oracle, m, n, k, da, db, out)


@pytest.mark.parametrize("da,db", [(E4, E4), (E5, E4), (F4, F4)])
def test_mfma_scaled_with_equal_scales_meets_the_f32_bound(client, oracle, da, db):
    run_scaled(client, oracle, 512, 256, 1024, da, db, ElemType.F32, scale_lo=127, scale_hi=128)     # all scales 2^0
    run_scaled(client, oracle, 256, 512, 512, da, db, ElemType.F32, scale_lo=131, scale_hi=132)      # all scales 2^4


@pytest.mark.parametrize("da", [E4, F4])
def test_mfma_scaled_single_and_odd_k_tiles_batches_and_broadcast(client, oracle, da):
    kt = 256 if da == F4 else 128
    for k in (kt, 2 * kt, 3 * kt, 5 * kt):                       # 1, 2, 3, 5 K-tiles: every prologue / tail combination
        run_scaled(client, oracle, 256, 256, k, da, da, ElemType.F32, seed=k)
    run_scaled(client, oracle, 256, 384, 2 * kt, da, da, ElemType.F32, batch=3)
    run_scaled(client, oracle, 512, 256, 2 * kt, da, da, ElemType.BF16, batch=2, bcast_b=True)


@pytest.mark.parametrize("da", [E4, F4])
def test_mfma_scaled_each_lane_uses_its_own_scale(client, oracle, da):
    """All-ones operands with a DIFFERENT scale for every (row, block): C[i][j] = 32 * sum_blk 2^(sa[i][blk]-127) 2^(sb[j][blk]-127)
    exactly -- any mix-up of lane <-> scale byte shows as a wrong power of two."""
    m = n = 256
    k = 512
    nb = k // 32
    one = 0x22 if da == F4 else 0x38                             # packed (1.0, 1.0) / e4m3 1.0
    a = np.full((m, k // (2 if da == F4 else 1)), one, dtype=np.uint8)
    sa = (124 + (np.arange(m)[:, None] * 3 + np.arange(nb)[None, :] * 5) % 7).astype(np.uint8)
    sb = (125 + (np.arange(n)[:, None] * 7 + np.arange(nb)[None, :] * 2) % 5).astype(np.uint8)
    c = TensorHandle.new_contiguous((m, n), client.empty(m * n * 4), ElemType.F32)
    ops.matmul_scaled(client, TensorHandle.from_numpy(client, a, da), TensorHandle.from_numpy(client, sa, ElemType.UE8M0),
                      TensorHandle.from_numpy(client, a, da), TensorHandle.from_numpy(client, sb, ElemType.UE8M0), c,
                      algo=N.GEMM_ALGO_LP_256W4)
    want = 32.0 * (2.0 ** (sa.astype(np.float64) - 127)) @ (2.0 ** (sb.astype(np.float64) - 127)).T
    assert np.array_equal(c.to_numpy(client), want.astype(np.float32))
    # a NaN scale poisons exactly its row
    sa2 = sa.copy()
    sa2[77, 3] = 0xFF
    ops.matmul_scaled(client, TensorHandle.from_numpy(client, a, da), TensorHandle.from_numpy(client, sa2, ElemType.UE8M0),
                      TensorHandle.from_numpy(client, a, da), TensorHandle.from_numpy(client, sb, ElemType.UE8M0), c,
                      algo=N.GEMM_ALGO_LP_256W4)
    got = c.to_numpy(client)
    assert np.all(np.isnan(got[77])) and np.array_equal(np.delete(got, 77, axis=0), np.delete(want.astype(np.float32), 77, axis=0))


def test_scaled_selection_and_errors(client):
    base = dict(m=512, n=512, k=512, batch=1, lda=512, ldb=512, ldc=512, ld_sa=16, ld_sb=16, dtype_a=N.DTYPE_F8E4M3,
                dtype_b=N.DTYPE_F8E4M3, dtype_c=N.DTYPE_F32, block=32)
    sel = lambda **kw: ops.gemm_scaled_select(client, N.GemmScaledDesc(**{**base, **kw}))
    assert sel() == N.GEMM_ALGO_LP_256W4
    assert sel(block=16, ld_sa=32, ld_sb=32) == N.GEMM_ALGO_GENERIC          # the hardware block is 32
    assert sel(k=192, lda=192, ldb=192, ld_sa=6, ld_sb=6) == N.GEMM_ALGO_GENERIC     # K not a multiple of the K-tile
    assert sel(dtype_c=N.DTYPE_F16) == N.GEMM_ALGO_GENERIC
    assert sel(dtype_a=N.DTYPE_F4E2M1X2, dtype_b=N.DTYPE_F4E2M1X2) == N.GEMM_ALGO_LP_256W4
    assert sel(dtype_a=N.DTYPE_F4E2M1X2, dtype_b=N.DTYPE_F4E2M1X2, k=384, lda=384, ldb=384, ld_sa=12, ld_sb=12) == N.GEMM_ALGO_GENERIC
    buf = client.empty(1 << 20)
    p = C.c_void_p(buf.device_ptr())

    def call(**kw):
        d = N.GemmScaledDesc(**{**base, **kw})
        client._s.check(client.lib.mi355_gemm_scaled(client.ctx, None, C.byref(d), p, p, p, p, p))
    for bad, code in ((dict(dtype_a=N.DTYPE_BF16), N.E_UNSUPPORTED), (dict(dtype_b=N.DTYPE_F4E2M1X2), N.E_UNSUPPORTED),
                      (dict(block=0), N.E_INVALID_ARGUMENT), (dict(block=48), N.E_INVALID_ARGUMENT), (dict(ld_sa=8), N.E_UNSUPPORTED_STRIDES),
                      (dict(lda=100), N.E_UNSUPPORTED_STRIDES), (dict(dtype_c=N.DTYPE_F8E4M3), N.E_UNSUPPORTED),
                      (dict(algo=N.GEMM_ALGO_LP_256W4, block=64, ld_sa=8, ld_sb=8), N.E_UNSUPPORTED)):
        with pytest.raises(ServerError) as e:
            call(**bad)
        assert e.value.code == code, bad
