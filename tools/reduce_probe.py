#!/usr/bin/env python3
"""Times the 1 GiB array-wide reductions (15 samples each, median / min); the command the reduce rocprof / PMC passes
of tools/gpu_check.sh and tools/pmc_all.sh profile."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cubecl_amd import Mi355Runtime, TensorHandle, ElemType
cl = Mi355Runtime.client(); lib, ctx = cl.lib, cl.ctx
n = 1 << 28
x = TensorHandle.uniform(cl, (n,), ElemType.F32, 1, 300, 0.0, 1.0)
ws = cl.empty(1 << 17); outs = cl.empty(64)
ea, eb = C.c_void_p(), C.c_void_p(); lib.mi355_event_create(ctx, C.byref(ea)); lib.mi355_event_create(ctx, C.byref(eb))
p_in, p_ws = C.c_void_p(x.device_ptr()), C.c_void_p(ws.device_ptr())
p_sum, p_val, p_idx = (C.c_void_p(outs.device_ptr() + o) for o in (0, 8, 16))
fns = {"sum": lambda: lib.mi355_reduce_sum_f32(ctx, None, p_in, n, p_sum, p_ws, ws.size),
       "argmax": lambda: lib.mi355_argmax_f32(ctx, None, p_in, n, p_val, p_idx, p_ws, ws.size),
       "fused": lambda: lib.mi355_sum_argmax_f32(ctx, None, p_in, n, p_sum, p_val, p_idx, p_ws, ws.size)}
for name, fn in fns.items():
    for _ in range(5): fn()
    cl.sync(); t = []
    for _ in range(15):
        lib.mi355_event_record(ctx, ea, None); fn(); lib.mi355_event_record(ctx, eb, None); lib.mi355_event_sync(ctx, eb)
        ms = C.c_float(); lib.mi355_event_elapsed_ms(ctx, ea, eb, C.byref(ms)); t.append(ms.value)
    t.sort()
    print(f"{name:7s} median {t[7]*1e3:7.1f} us ({n*4/t[7]/1e6:7.1f} GB/s)  min {t[0]*1e3:7.1f} us ({n*4/t[0]/1e6:7.1f} GB/s)", flush=True)
