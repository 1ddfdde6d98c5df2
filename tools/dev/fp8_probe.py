#!/usr/bin/env python3
"""Dev (GPU box): fp8 e4m3 and MXFP8 GEMM timings of whatever library MI355CUBE_LIB names (A/B of kernel variants: run the two
libraries alternately).  usage: [MI355CUBE_LIB=...] python tools/dev/fp8_probe.py [label]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from cubecl_amd import ElemType, Mi355Runtime, TensorHandle
from cubecl_amd import _native as N
label = sys.argv[1] if len(sys.argv) > 1 else "lib"
cl = Mi355Runtime.client(); ev = bench.Events(cl); lib, ctx = cl.lib, cl.ctx
res = []
for S in (8192, 4096, 6144):
    a = TensorHandle.uniform(cl, (S, S), ElemType.F8E4M3, 7, 900, -1.0, 1.0)
    b = TensorHandle.uniform(cl, (S, S), ElemType.F8E4M3, 7, 901, -1.0, 1.0)
    c = cl.empty(S * S * 2)
    d = N.GemmDesc(m=S, n=S, k=S, batch=1, lda=S, ldb=S, ldc=S, dtype_ab=N.DTYPE_F8E4M3, dtype_c=N.DTYPE_BF16, trans_b=1, algo=0)
    call = lambda: cl._s.check(lib.mi355_gemm(ctx, None, C.byref(d), a.device_ptr(), b.device_ptr(), c.device_ptr()))
    bench.time_op(cl, ev, call, 40)
    t = min(bench.time_op(cl, ev, call, 30) for _ in range(3))
    res.append(f"fp8 {S}^3 {2.0 * S ** 3 / t / 1e9:7.1f} TF")
S = 8192
a = TensorHandle.uniform(cl, (S, S), ElemType.F8E4M3, 7, 910, -1.0, 1.0)
b = TensorHandle.uniform(cl, (S, S), ElemType.F8E4M3, 7, 911, -1.0, 1.0)
sa = TensorHandle.uniform(cl, (S * S // 32,), ElemType.UE8M0, 7, 912, 124.0, 131.0)
sb = TensorHandle.uniform(cl, (S * S // 32,), ElemType.UE8M0, 7, 913, 124.0, 131.0)
c = cl.empty(S * S * 2)
d = N.GemmScaledDesc(m=S, n=S, k=S, batch=1, lda=S, ldb=S, ldc=S, ld_sa=S // 32, ld_sb=S // 32, dtype_a=int(ElemType.F8E4M3), dtype_b=int(ElemType.F8E4M3),
                     dtype_c=N.DTYPE_BF16, block=32)
call = lambda: cl._s.check(lib.mi355_gemm_scaled(ctx, None, C.byref(d), a.device_ptr(), sa.device_ptr(), b.device_ptr(), sb.device_ptr(), c.device_ptr()))
bench.time_op(cl, ev, call, 40)
t = min(bench.time_op(cl, ev, call, 30) for _ in range(3))
res.append(f"mxfp8 {S}^3 {2.0 * S ** 3 / t / 1e9:7.1f} TF")
print(f"{label:10s} " + "   ".join(res), flush=True)
