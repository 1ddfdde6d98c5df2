"""Seeded random descriptors through `mi355_gemm` with GEMM_ALGO_AUTO, against an independent f64 product.

The hand-picked cases of test_gpu_gemm.py sit ON the dispatcher's thresholds; this walks between them: dimensions drawn around
every tile size the ten kernels use (1, 16, 32, 64, 128, 256 and their neighbours), ragged and padded leading dimensions,
batches, a broadcast rhs, both rhs layouts (`MatrixBatchLayout::Contiguous` and the transposed form,
crates/cubecl-std/src/tensor/matrix_batch_layout.rs:21-79), every dtype pair the entry point takes.  Semantics and tolerance are
run_case()'s (crates/cubecl-core/src/runtime_tests/cmma.rs:695-722, :1160-1177; 1e-5 of sum |a||b| for f32 outputs, one unit in
the last place on top of it for 16-bit outputs).  The draw is a pure function of the seed, so a failure names its case.
"""
import numpy as np
import pytest

from cubecl_amd import ElemType
from cubecl_amd import _native as N
from test_gpu_gemm import run_case

pytestmark = pytest.mark.gpu
import os
OFFSET = int(os.environ.get("MI355_FUZZ_OFFSET", "0"))     # soak runs: the same tests over another stretch of seeds

EDGES = [1, 2, 3, 5, 8, 15, 16, 17, 24, 31, 32, 33, 48, 63, 64, 65, 96, 100, 127, 128, 129, 160, 192, 255, 256, 257, 320, 384, 500,
         512, 513, 640, 768, 1000, 1024, 1025, 1536, 2048, 2049, 3072, 4096]
K_EDGES = [1, 7, 8, 16, 31, 32, 40, 63, 64, 65, 96, 128, 130, 192, 256, 264, 320, 512, 576, 1000, 1024, 2048, 2560, 3072, 4096, 8192]
PAIRS = [(ElemType.BF16, ElemType.BF16), (ElemType.BF16, ElemType.F32), (ElemType.F16, ElemType.F16), (ElemType.F16, ElemType.F32),
         (ElemType.F32, ElemType.F32), (ElemType.F8E4M3, ElemType.BF16), (ElemType.F8E5M2, ElemType.F32)]
WORK = 1.0e9        # multiply-adds per case: bounds the f64 product on the host


def draw(seed):
    rng = np.random.default_rng(0x5EEDC0BE + seed + OFFSET)
    kind = rng.integers(0, 5)
    for _ in range(1000):
        if kind == 0:      # anything
            m, n, k = (int(rng.choice(EDGES)), int(rng.choice(EDGES)), int(rng.choice(K_EDGES)))
        elif kind == 1:    # skinny: few rows or columns, long K
            s, big = int(rng.choice([1, 2, 3, 4, 8, 15, 16, 17, 32, 33, 48, 64, 65])), int(rng.choice([256, 512, 1000, 1024, 2048, 4096]))
            k = int(rng.choice([512, 1024, 2048, 4096, 8192]))
            m, n = (s, big) if rng.integers(0, 2) else (big, s)
        elif kind == 2:    # few tiles, long K: the split-K band
            m, n, k = int(rng.choice([64, 96, 128, 192, 256, 384, 512])), int(rng.choice([64, 128, 200, 256, 512, 768])), int(rng.choice([2048, 4096, 8192, 16384]))
        elif kind == 3:    # many tiles, short K: the output-bound band
            m, n, k = int(rng.choice([512, 1024, 1100, 2048, 4096])), int(rng.choice([512, 1024, 1536, 2048, 4096])), int(rng.choice([8, 16, 32, 64, 96, 128, 192, 256]))
        else:              # whole tiles of the 256 / 128 kernels, off by one block here and there
            m = 128 * int(rng.integers(1, 17)) + int(rng.choice([0, 0, 0, 1, -1, 32]))
            n = 128 * int(rng.integers(1, 17)) + int(rng.choice([0, 0, 0, 8, -8, 64]))
            k = 64 * int(rng.integers(1, 33)) + int(rng.choice([0, 0, 0, 8, 32]))
        batch = int(rng.choice([1, 1, 1, 2, 3, 5]))
        if m * n * k * batch <= WORK and m * n * batch <= 2.5e7:
            break
    dtype, out = PAIRS[int(rng.integers(0, len(PAIRS)))]
    trans_b = bool(rng.integers(0, 2))
    pad = lambda: int(rng.choice([0, 0, 0, 8, 16, 24, 64, 1, 3]))
    kw = dict(lda=k + pad(), ldb=(k if trans_b else n) + pad(), ldc=n + pad(), batch=batch, bcast_b=bool(batch > 1 and rng.integers(0, 3) == 0))
    return m, n, k, dtype, out, trans_b, kw


def draw_transposed_a(seed):
    """A stored [K][M] (MatrixBatchLayout::MildlyPermuted { transposed: true } on the lhs, matrix_batch_layout.rs:21-79): the same
    draw, with lda re-drawn around M; together with a row-major B most 16-bit cases land on the 128x128 kernel's native form
    (gemm_lp128.hip ATN), the rest on the re-layout pass."""
    m, n, k, dtype, out, trans_b, kw = draw(5000 + seed)
    rng = np.random.default_rng(0x7A + seed + OFFSET)
    if rng.integers(0, 3):
        trans_b = False                                     # lhs^T . grad_out: both operands walked along their rows by K
        kw["ldb"] = n + int(rng.choice([0, 0, 8, 16]))
    if rng.integers(0, 2) and m > 8:
        m = m // 8 * 8                                      # the native form wants whole 16-byte pieces of A's rows
    kw["lda"] = m + int(rng.choice([0, 0, 0, 8, 16, 24, 1]))
    return m, n, k, dtype, out, trans_b, kw


def draw_big(seed):
    """More than 128 tiles of 256^2 (or many of 128^2), so that the persistent and the 256-tile kernels answer: 12 cases, a few
    seconds of f64 product each on the host."""
    rng = np.random.default_rng(0xB16C0BE + seed + OFFSET)
    many = seed % 3 == 2            # more than 256 tiles of 256^2, 16-bit: the persistent kernels' ground
    while True:
        m = 256 * int(rng.integers(16 if many else 10, 21 if many else 19)) + int(rng.choice([0, 0, 0, -8, 32, 128]))
        n = 256 * int(rng.integers(16 if many else 10, 21 if many else 19)) + int(rng.choice([0, 0, 0, -8, 64, 128]))
        k = 64 * int(rng.integers(3, 12 if many else 20)) + int(rng.choice([0, 0, 0, 8, 32]))
        batch = 1 if many else int(rng.choice([1, 1, 2]))
        if m * n * k * batch <= 2.4e10:
            break
    dtype, out = PAIRS[int(rng.integers(0, 4 if many else len(PAIRS)))]
    trans_b = bool(rng.integers(0, 2))
    pad = lambda: int(rng.choice([0, 0, 8, 64]))
    return m, n, k, dtype, out, trans_b, dict(lda=k + pad(), ldb=(k if trans_b else n) + pad(), ldc=n + pad(), batch=batch, bcast_b=False)


@pytest.mark.parametrize("seed", range(160))
def test_auto_dispatch_on_random_descriptors(client, oracle, seed):
    m, n, k, dtype, out, trans_b, kw = draw(seed)
    run_case(client, oracle, m, n, k, dtype, out, trans_b, N.GEMM_ALGO_AUTO, seed_t=1000 + seed, **kw)


@pytest.mark.parametrize("seed", range(80))
def test_auto_dispatch_on_random_descriptors_with_a_transposed_lhs(client, oracle, seed):
    m, n, k, dtype, out, trans_b, kw = draw_transposed_a(seed)
    run_case(client, oracle, m, n, k, dtype, out, trans_b, N.GEMM_ALGO_AUTO, seed_t=3000 + seed, trans_a=True, **kw)


@pytest.mark.parametrize("seed", range(12))
def test_auto_dispatch_on_random_large_descriptors(client, oracle, seed):
    m, n, k, dtype, out, trans_b, kw = draw_big(seed)
    run_case(client, oracle, m, n, k, dtype, out, trans_b, N.GEMM_ALGO_AUTO, seed_t=2000 + seed, **kw)


@pytest.mark.skipif(OFFSET != 0, reason="the coverage claim is about the committed stretch of seeds")
def test_the_draws_reach_the_kernels(client):
    """The fuzz means little if AUTO answers every case with the same kernel."""
    import ctypes as C
    chosen = {}
    for fn, count in ((draw, 160), (draw_big, 12)):
        for seed in range(count):
            m, n, k, dtype, out, trans_b, kw = fn(seed)
            d = N.GemmDesc(m=m, n=n, k=k, batch=kw["batch"], lda=kw["lda"], ldb=kw["ldb"], ldc=kw["ldc"], stride_a=m * kw["lda"],
                           stride_b=0 if kw["bcast_b"] else (n if trans_b else k) * kw["ldb"], stride_c=m * kw["ldc"],
                           dtype_ab=int(dtype), dtype_c=int(out), trans_a=0, trans_b=int(trans_b), algo=N.GEMM_ALGO_AUTO)
            algo = C.c_int32(-1)
            client._s.check(client.lib.mi355_gemm_select(client.ctx, C.byref(d), C.byref(algo)))
            chosen[algo.value] = chosen.get(algo.value, 0) + 1
    print("kernels chosen over the draws:", sorted(chosen.items()))
    want = {N.GEMM_ALGO_GENERIC, N.GEMM_ALGO_F32_MFMA, N.GEMM_ALGO_LP_128, N.GEMM_ALGO_LP_256W4, N.GEMM_ALGO_SKINNY, N.GEMM_ALGO_STREAM64}
    assert want <= set(chosen), sorted(chosen.items())
    assert set(chosen) & {N.GEMM_ALGO_LP_256P, N.GEMM_ALGO_LP_256Q, N.GEMM_ALGO_LP_256QM, N.GEMM_ALGO_LP_256X128}, sorted(chosen.items())
