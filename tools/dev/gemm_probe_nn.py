#!/usr/bin/env python3
"""Dev: bf16 GEMM with row-major B (re-layout path) vs [N][K] B."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cubecl_amd import Mi355Runtime, TensorHandle, ElemType
from cubecl_amd import _native as N
cl = Mi355Runtime.client(); lib, ctx = cl.lib, cl.ctx
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
a = TensorHandle.uniform(cl, (S * S,), ElemType.BF16, 1, 1, -1.0, 1.0); b = TensorHandle.uniform(cl, (S * S,), ElemType.BF16, 1, 2, -1.0, 1.0)
c = cl.empty(S * S * 2)
ea, eb = C.c_void_p(), C.c_void_p(); lib.mi355_event_create(ctx, C.byref(ea)); lib.mi355_event_create(ctx, C.byref(eb))
for tb in (1, 0):
    d = N.GemmDesc(m=S, n=S, k=S, batch=1, lda=S, ldb=S, ldc=S, stride_a=S*S, stride_b=S*S, stride_c=S*S, dtype_ab=N.DTYPE_BF16, dtype_c=N.DTYPE_BF16, trans_a=0, trans_b=tb, algo=0)
    fn = lambda: lib.mi355_gemm(ctx, None, C.byref(d), C.c_void_p(a.device_ptr()), C.c_void_p(b.device_ptr()), C.c_void_p(c.device_ptr()))
    for _ in range(80): fn()
    cl.sync(); lib.mi355_event_record(ctx, ea, None)
    for _ in range(30): fn()
    lib.mi355_event_record(ctx, eb, None); lib.mi355_event_sync(ctx, eb)
    ms = C.c_float(); lib.mi355_event_elapsed_ms(ctx, ea, eb, C.byref(ms))
    print(f"bf16 {S}^3 trans_b={tb}: {ms.value/30:.4f} ms  {2.0*S**3/(ms.value/30)/1e9:.0f} TF (steady state)", flush=True)
