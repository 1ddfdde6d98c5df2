#!/usr/bin/env python3
"""Dev: time the f32 4096^3 GEMM (NT and NN). usage: python tools/dev/gemm_probe_f32.py [size]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cubecl_amd import Mi355Runtime, TensorHandle, ElemType
from cubecl_amd import _native as N
cl = Mi355Runtime.client(); lib, ctx = cl.lib, cl.ctx
M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
a = TensorHandle.uniform(cl, (M, M), ElemType.F32, 1, 400, -1.0, 1.0)
b = TensorHandle.uniform(cl, (M, M), ElemType.F32, 1, 401, -1.0, 1.0)
c = cl.empty(M * M * 4)
ea, eb = C.c_void_p(), C.c_void_p(); lib.mi355_event_create(ctx, C.byref(ea)); lib.mi355_event_create(ctx, C.byref(eb))
for name, tb in (("NT", 1), ("NN", 0)):
    d = N.GemmDesc(m=M, n=M, k=M, batch=1, lda=M, ldb=M, ldc=M, stride_a=M*M, stride_b=M*M, stride_c=M*M,
                   dtype_ab=N.DTYPE_F32, dtype_c=N.DTYPE_F32, trans_a=0, trans_b=tb, algo=0)
    fn = lambda: lib.mi355_gemm(ctx, None, C.byref(d), C.c_void_p(a.device_ptr()), C.c_void_p(b.device_ptr()), C.c_void_p(c.device_ptr()))
    for _ in range(3): fn()
    cl.sync(); t = []
    for _ in range(10):
        lib.mi355_event_record(ctx, ea, None); fn(); lib.mi355_event_record(ctx, eb, None); lib.mi355_event_sync(ctx, eb)
        ms = C.c_float(); lib.mi355_event_elapsed_ms(ctx, ea, eb, C.byref(ms)); t.append(ms.value)
    t.sort()
    lib.mi355_event_record(ctx, ea, None)
    for _ in range(10): fn()
    lib.mi355_event_record(ctx, eb, None); lib.mi355_event_sync(ctx, eb)
    ms = C.c_float(); lib.mi355_event_elapsed_ms(ctx, ea, eb, C.byref(ms)); b2b = ms.value / 10
    print(f"f32 {name} {M}^3: median {t[5]:.4f} ms {2.0*M**3/t[5]/1e9:6.1f} TF   min {t[0]:.4f} ms {2.0*M**3/t[0]/1e9:6.1f} TF   back-to-back x10 {b2b:.4f} ms {2.0*M**3/b2b/1e9:6.1f} TF", flush=True)
