#!/usr/bin/env python3
"""Dev: which columns of a forced gemm_nnrows launch differ from numpy, and is the set stable run to run."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cubecl_amd import ElemType, Mi355Runtime, TensorHandle
from cubecl_amd import _native as N
cl = Mi355Runtime.client(); lib, ctx = cl.lib, cl.ctx
m, n, k = [int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (2, 131072, 512))]
rng = np.random.default_rng(1)
x = rng.integers(-2, 3, size=(m, k)).astype(np.float32)
w = rng.integers(-2, 3, size=(k, n)).astype(np.float32)
def bf(a):
    return (a.view(np.uint32) >> 16).astype(np.uint16)
xa = cl.create_from_slice(bf(x)); wb = cl.create_from_slice(bf(w))
ref = x @ w
d = N.GemmDesc(m=m, n=n, k=k, batch=1, lda=k, ldb=n, ldc=n, stride_a=m * k, stride_b=k * n, stride_c=m * n, dtype_ab=N.DTYPE_BF16, dtype_c=N.DTYPE_F32,
               trans_a=0, trans_b=0, algo=N.GEMM_ALGO_NNROWS)
prev = None
for run in range(4):
    c = cl.empty(m * n * 4)
    rc = lib.mi355_gemm(ctx, None, C.byref(d), xa.device_ptr(), wb.device_ptr(), c.device_ptr())
    got = cl.read_one(c).view(np.float32).reshape(m, n)
    bad = np.argwhere(got != ref)
    cols = np.unique(bad[:, 1])
    print(f"run {run}: rc {rc} bad elements {len(bad)} bad columns {len(cols)}; strips {np.unique(cols // 512)[:20]} ...; col % 512 // 8 (lane) {np.unique((cols % 512) // 8)[:70]}")
    if len(bad):
        i, j = bad[0]
        # which 4-row groups are off for the first bad column: contribution of every group
        contrib = (x[i].reshape(-1, 4) * w[:, j].reshape(-1, 4)).sum(axis=1)
        print("   first bad", (int(i), int(j)), "got", got[i, j], "ref", ref[i, j], "diff", got[i, j] - ref[i, j])
        print("   same bad set as previous run:", prev is not None and np.array_equal(prev, bad))
    prev = bad
