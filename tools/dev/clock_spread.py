#!/usr/bin/env python3
"""Dev: per-XCD shader clock over 30 back-to-back 8192^3 bf16 GEMMs (s_memtime / s_memrealtime samples)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from cubecl_amd import Mi355Runtime, TensorHandle, ElemType
from cubecl_amd import _native as N
cl = Mi355Runtime.client(); lib, ctx = cl.lib, cl.ctx
S = 8192
a = TensorHandle.uniform(cl, (S, S), ElemType.BF16, 1, 1, -1.0, 1.0); b = TensorHandle.uniform(cl, (S, S), ElemType.BF16, 1, 2, -1.0, 1.0)
c = cl.empty(S * S * 2)
d = N.GemmDesc(m=S, n=S, k=S, batch=1, lda=S, ldb=S, ldc=S, stride_a=S*S, stride_b=S*S, stride_c=S*S, dtype_ab=N.DTYPE_BF16, dtype_c=N.DTYPE_BF16, trans_a=0, trans_b=1, algo=0)
clk = cl.empty(256)
ea, eb = C.c_void_p(), C.c_void_p(); lib.mi355_event_create(ctx, C.byref(ea)); lib.mi355_event_create(ctx, C.byref(eb))
for rep in range(4):
    lib.mi355_memset(ctx, None, C.c_void_p(clk.device_ptr()), 0, 256)
    for _ in range(5): lib.mi355_gemm(ctx, None, C.byref(d), C.c_void_p(a.device_ptr()), C.c_void_p(b.device_ptr()), C.c_void_p(c.device_ptr()))
    cl.sync()
    lib.mi355_probe_clock(ctx, None, C.c_void_p(clk.device_ptr()))
    lib.mi355_event_record(ctx, ea, None)
    for _ in range(30): lib.mi355_gemm(ctx, None, C.byref(d), C.c_void_p(a.device_ptr()), C.c_void_p(b.device_ptr()), C.c_void_p(c.device_ptr()))
    lib.mi355_event_record(ctx, eb, None)
    lib.mi355_probe_clock(ctx, None, C.c_void_p(clk.device_ptr() + 128))
    lib.mi355_event_sync(ctx, eb); cl.sync()
    ms = C.c_float(); lib.mi355_event_elapsed_ms(ctx, ea, eb, C.byref(ms))
    t = cl.read_one(clk).view(np.uint64).reshape(2, 8, 2).astype(np.float64)
    ghz = [(t[1, x, 0] - t[0, x, 0]) / max(t[1, x, 1] - t[0, x, 1], 1) * 0.1 for x in range(8)]
    span = [(t[1, x, 1] - t[0, x, 1]) / 100e3 for x in range(8)]
    print(f"rep {rep}: {ms.value/30:.4f} ms/GEMM  {2.0*S**3/(ms.value/30)/1e9:.0f} TF  clocks GHz {[round(g,3) for g in ghz]} mean {np.mean(ghz):.3f}  spans ms {[round(s,2) for s in span]}")
