"""Is v_mfma_f32_32x32x64_f8f6f4 exact?  Compares the tile kernels with the generic kernel (an f32 fma chain: exact products,
f32 accumulation) and an f64 product, on random and on constructed operands (one large product plus many tiny ones)."""
import sys
import numpy as np
sys.path.insert(0, ".")
import oracle
from cubecl_amd import ElemType, TensorHandle, ops, Mi355Runtime
from cubecl_amd import _native as N

oracle.lib()
client = Mi355Runtime.client()


def run(abits, bbits, m, n, k, dtype, algo):
    ta = TensorHandle.new(client.create_from_slice(abits), (m, k), (k, 1), dtype)
    tb = TensorHandle.new(client.create_from_slice(bbits), (k, n), (1, k), dtype)
    c = TensorHandle.new_contiguous((m, n), client.empty(m * n * 4), ElemType.F32)
    ops.matmul(client, ta, tb, c, algo=algo)
    return c.to_numpy(client).reshape(m, n).astype(np.float64)


for dtype in (ElemType.F8E5M2, ElemType.F8E4M3):
    for k in (128,):
        m = n = 256
        a = oracle.to_fp8(oracle.fill_uniform(m * k, 5, -1.0, 1.0), int(dtype)).reshape(m, k)
        b = oracle.to_fp8(oracle.fill_uniform(n * k, 6, -1.0, 1.0), int(dtype)).reshape(n, k)
        A = oracle.from_fp8(a, int(dtype)).astype(np.float64)
        B = oracle.from_fp8(b, int(dtype)).astype(np.float64)
        ref = A @ B.T
        bound = np.abs(A) @ np.abs(B.T)
        line = [f"{dtype.name} k={k:5d}"]
        for name, algo in (("generic", N.GEMM_ALGO_GENERIC), ("lp128", N.GEMM_ALGO_LP_128), ("lp256w4", N.GEMM_ALGO_LP_256W4), ("auto", N.GEMM_ALGO_AUTO)):
            try:
                got = run(a, b, m, n, k, dtype, algo)
                err = np.abs(got - ref)
                line.append(f"{name}: max|err| {err.max():.3e} err/bound {np.max(err / (bound + 1e-30)):.2e} nonzero {np.count_nonzero(err)}")
            except Exception as e:
                line.append(f"{name}: {type(e).__name__}")
        print(" | ".join(line), flush=True)
    # constructed: a[0][0] * b[0][0] = 1 and 63 products of 2^-t each
    for t in (4, 6, 8, 10, 12, 13, 14, 15, 16, 18, 20, 22, 24):
        k = 128
        one = 0x3C if dtype == ElemType.F8E5M2 else 0x38
        av = np.zeros((256, k), np.float32); bv = np.zeros((256, k), np.float32)
        av[:, 0] = 1.0; bv[:, 0] = 1.0
        half = t // 2
        av[:, 1:64] = 2.0 ** -half; bv[:, 1:64] = 2.0 ** -(t - half)
        a = oracle.to_fp8(av.reshape(-1), int(dtype)).reshape(256, k); b = oracle.to_fp8(bv.reshape(-1), int(dtype)).reshape(256, k)
        A = oracle.from_fp8(a, int(dtype)).astype(np.float64); B = oracle.from_fp8(b, int(dtype)).astype(np.float64)
        ref = (A @ B.T)[0, 0]
        got = run(a, b, 256, 256, k, dtype, N.GEMM_ALGO_LP_128)[0, 0]
        # the same tiny products WITHOUT the large one: are they dropped or only truncated against it?
        a2 = a.copy(); a2[:, 0] = 0
        alone = run(a2, b, 256, 256, k, dtype, N.GEMM_ALGO_LP_128)[0, 0]
        gen = run(a, b, 256, 256, k, dtype, N.GEMM_ALGO_GENERIC)[0, 0]
        print(f"{dtype.name} 1 + 63 x 2^-{t}: exact {ref!r} mfma {got!r} generic {gen!r}; tiny ones alone: {alone!r} (exact {ref - 1.0!r})", flush=True)
