#!/usr/bin/env python3
"""Dev: steady-state TFLOP/s (median of 30-launch blocks after the DVFS ramp) for algo x shape.
usage: python tools/dev/sustain2.py "5 6" "dtype:m,n,k[,batch]" ..."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cubecl_amd import Mi355Runtime, TensorHandle, ElemType
from cubecl_amd import _native as N
cl = Mi355Runtime.client(); lib, ctx = cl.lib, cl.ctx
algos = [int(x) for x in sys.argv[1].split()]
for spec in sys.argv[2:]:
    dt, dims = spec.split(":")
    v = [int(x) for x in dims.split(",")]
    m, n, k = v[:3]; batch = v[3] if len(v) > 3 else 1
    et, code = (ElemType.F32, N.DTYPE_F32) if dt == "f32" else (ElemType.BF16, N.DTYPE_BF16)
    a = TensorHandle.uniform(cl, (batch * m * k,), et, 1, 1, -1.0, 1.0); b = TensorHandle.uniform(cl, (batch * n * k,), et, 1, 2, -1.0, 1.0)
    c = cl.empty(batch * m * n * et.size())
    flop = 2.0 * m * n * k * batch
    per = max(3, int(30 * 0.8e-3 / (flop / 1.3e15)))      # launches per block: ~25 ms
    res = {}
    for rep in range(2):
        for algo in algos:
            d = N.GemmDesc(m=m, n=n, k=k, batch=batch, lda=k, ldb=k, ldc=n, stride_a=m*k, stride_b=n*k, stride_c=m*n, dtype_ab=code, dtype_c=code, trans_a=0, trans_b=1, algo=algo)
            NB = 24
            evs = [C.c_void_p() for _ in range(NB + 1)]
            for e in evs: lib.mi355_event_create(ctx, C.byref(e))
            cl.sync(); lib.mi355_event_record(ctx, evs[0], None)
            for i in range(NB):
                for _ in range(per): lib.mi355_gemm(ctx, None, C.byref(d), C.c_void_p(a.device_ptr()), C.c_void_p(b.device_ptr()), C.c_void_p(c.device_ptr()))
                lib.mi355_event_record(ctx, evs[i + 1], None)
            cl.sync(); out = []
            for i in range(NB):
                ms = C.c_float(); lib.mi355_event_elapsed_ms(ctx, evs[i], evs[i + 1], C.byref(ms)); out.append(flop * per / ms.value / 1e9)
            for e in evs: lib.mi355_event_destroy(ctx, e)
            s = sorted(out[6:]); res.setdefault(algo, []).append(s[len(s) // 2])
    print(spec, {a_: [round(x, 1) for x in r] for a_, r in res.items()}, flush=True)
    del a, b, c; cl.flush()
