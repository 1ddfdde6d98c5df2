#!/usr/bin/env python3
"""Dev: per-tile loop / epilogue cycles of the persistent lp256p kernel (needs a -DP_TRACE variant build)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from cubecl_amd import Mi355Runtime, TensorHandle, ElemType
from cubecl_amd import _native as N
cl = Mi355Runtime.client(); lib, ctx = cl.lib, cl.ctx
for spec in sys.argv[1:]:
    v = [int(x) for x in spec.split(",")]; m, n, k = v[:3]; batch = v[3] if len(v) > 3 else 1
    a = TensorHandle.uniform(cl, (batch * m * k,), ElemType.BF16, 1, 1, -1.0, 1.0); b = TensorHandle.uniform(cl, (batch * n * k,), ElemType.BF16, 1, 2, -1.0, 1.0)
    c = cl.empty(batch * m * n * 2)
    d = N.GemmDesc(m=m, n=n, k=k, batch=batch, lda=k, ldb=k, ldc=n, stride_a=m*k, stride_b=n*k, stride_c=m*n, dtype_ab=N.DTYPE_BF16, dtype_c=N.DTYPE_BF16, trans_a=0, trans_b=1, algo=6)
    for _ in range(60): lib.mi355_gemm(ctx, None, C.byref(d), C.c_void_p(a.device_ptr()), C.c_void_p(b.device_ptr()), C.c_void_p(c.device_ptr()))
    cl.sync()
    buf = np.zeros(256 * 16, dtype=np.uint64)
    lib.mi355_dev_p_trace(buf.ctypes.data_as(C.c_void_p))
    t = buf.reshape(256, 16).astype(np.float64)
    tiles = (m // 256) * (n // 256) * batch; per = min(7, tiles // 256)
    d_ = np.diff(t[:, :1 + 2 * per], axis=1)
    med = np.median(d_, axis=0)
    print(spec, "nk", k // 64, "per-WG segments (loop, epilogue, loop, epilogue, ...):", " ".join(f"{x:.0f}" for x in med), flush=True)
    del a, b, c; cl.flush()
