#!/usr/bin/env python3
"""Dev: where a workgroup of gemm_nnrows.hip spends its time (needs tools/dev/build_variants.sh nnrtrace "-DNNR_TRACE" gemm_nnrows.hip).
usage (GPU box): MI355CUBE_LIB=$PWD/cubecl_amd/csrc/variants/libmi355cube_nnrtrace.so python tools/dev/nnrows_trace.py M N K [M N K ...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from cubecl_amd import ElemType, Mi355Runtime, TensorHandle
from cubecl_amd import _native as N
cl = Mi355Runtime.client(); lib, ctx = cl.lib, cl.ctx
args = [int(x) for x in sys.argv[1:]] or [16, 8192, 8192]
for m, n, k in zip(args[0::3], args[1::3], args[2::3]):
    sets = [(TensorHandle.uniform(cl, (m, k), ElemType.BF16, 1, 2 * i + 1, -1.0, 1.0), TensorHandle.uniform(cl, (k, n), ElemType.BF16, 1, 2 * i + 2, -1.0, 1.0))
            for i in range(4)]
    c = cl.empty(m * n * 2)
    d = N.GemmDesc(m=m, n=n, k=k, batch=1, lda=k, ldb=n, ldc=n, dtype_ab=N.DTYPE_BF16, dtype_c=N.DTYPE_BF16, trans_a=0, trans_b=0, algo=N.GEMM_ALGO_NNROWS)
    buf = np.zeros(1024 * 8, dtype=np.uint64)
    for i in range(6):
        a, b = sets[i % 4]
        cl.sync()
        lib.mi355_dev_nnr_trace(buf.ctypes.data_as(C.c_void_p), 1)
        cl._s.check(lib.mi355_gemm(ctx, None, C.byref(d), a.device_ptr(), b.device_ptr(), c.device_ptr()))
        cl.sync()
    lib.mi355_dev_nnr_trace(buf.ctypes.data_as(C.c_void_p), 0)
    t = buf.reshape(1024, 8).astype(np.float64)
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    names = ["entry", "x staged", "K loop left", "LDS barrier", "partials out", "ticket back", "fold done (last of a strip)"]
    print(f"{m} x {n} x {k}: {len(t)} workgroups; s_memtime ticks (100 MHz: 1 tick = 10 ns), relative to the first workgroup's entry")
    for j, nm in enumerate(names):
        col = t[:, j][t[:, j] > 0]
        if len(col):
            rel = col - t0
            print(f"  {nm:28s} n={len(col):4d}  min {rel.min():7.0f}  median {np.median(rel):7.0f}  max {rel.max():7.0f}"
                  + (f"   (phase: median {np.median((t[:, j] - t[:, j - 1])[t[:, j] > 0]):6.0f})" if j else ""))
    del sets, c
    cl.memory_cleanup()
