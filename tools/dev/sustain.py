#!/usr/bin/env python3
"""Dev: TFLOP/s of consecutive blocks of 30 back-to-back 8192^3 bf16 GEMMs over ~1.5 s (DVFS ramp / steady state)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cubecl_amd import Mi355Runtime, TensorHandle, ElemType
from cubecl_amd import _native as N
cl = Mi355Runtime.client(); lib, ctx = cl.lib, cl.ctx
S = 8192
a = TensorHandle.uniform(cl, (S, S), ElemType.BF16, 1, 1, -1.0, 1.0); b = TensorHandle.uniform(cl, (S, S), ElemType.BF16, 1, 2, -1.0, 1.0)
c = cl.empty(S * S * 2)
d = N.GemmDesc(m=S, n=S, k=S, batch=1, lda=S, ldb=S, ldc=S, stride_a=S*S, stride_b=S*S, stride_c=S*S, dtype_ab=N.DTYPE_BF16, dtype_c=N.DTYPE_BF16, trans_a=0, trans_b=1, algo=int(sys.argv[1]) if len(sys.argv) > 1 else 0)
NB = 60
evs = [C.c_void_p() for _ in range(NB + 1)]
for e in evs: lib.mi355_event_create(ctx, C.byref(e))
cl.sync()
lib.mi355_event_record(ctx, evs[0], None)
for i in range(NB):
    for _ in range(30): lib.mi355_gemm(ctx, None, C.byref(d), C.c_void_p(a.device_ptr()), C.c_void_p(b.device_ptr()), C.c_void_p(c.device_ptr()))
    lib.mi355_event_record(ctx, evs[i + 1], None)
cl.sync()
out = []
for i in range(NB):
    ms = C.c_float(); lib.mi355_event_elapsed_ms(ctx, evs[i], evs[i + 1], C.byref(ms))
    out.append(2.0 * S**3 * 30 / ms.value / 1e9)
print("TF per block of 30:", " ".join(f"{x:.0f}" for x in out))
