#!/usr/bin/env python3
"""Dev: per-segment cycle totals of the lp256 kernel (needs a -DLP256_TRACE variant build).
usage (GPU box): MI355CUBE_LIB=.../libmi355cube_trace.so python tools/dev/lp256_trace.py [size]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from cubecl_amd import Mi355Runtime, TensorHandle, ElemType, ops, _native as N
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
cl = Mi355Runtime.client()
a = TensorHandle.uniform(cl, (S, S), ElemType.BF16, 1, 1, -1.0, 1.0)
b = TensorHandle.uniform(cl, (S, S), ElemType.BF16, 1, 2, -1.0, 1.0)
bt = TensorHandle.new(b.handle, (S, S), (1, S), ElemType.BF16)
c = TensorHandle.new_contiguous((S, S), cl.empty(S * S * 2), ElemType.BF16)
for _ in range(3):
    ops.matmul(cl, a, bt, c, algo=N.GEMM_ALGO_LP_256)
cl.sync()
buf = np.zeros(64 * 8 * 12, dtype=np.uint64)
rc = cl.lib.mi355_dev_lp256_trace(buf.ctypes.data_as(C.c_void_p))
t = buf.reshape(64, 8, 12).astype(np.float64)
nk = S // 64
names = ["L0", "bar", "C0", "bar", "L1", "bar", "C1(+vm4)", "bar"]
print(f"rc={rc} size={S} nk={nk}: mean cycles per K-tile segment (memtime ticks), group0 waves | group1 waves")
for k in range(8):
    g0 = t[:, :4, k].mean() / nk; g1 = t[:, 4:, k].mean() / nk
    print(f"  {names[k]:10s} g0 {g0:8.1f}   g1 {g1:8.1f}")
print("  total/K-tile g0 %.1f g1 %.1f" % (t[:, :4, :8].sum(axis=2).mean() / nk, t[:, 4:, :8].sum(axis=2).mean() / nk))
print('  per-tile cycles: prologue %.0f loop %.0f epilogue %.0f total %.0f' % tuple(t[:, :, 8 + i].mean() for i in range(4)))
import time
ev=[]
t0=time.perf_counter()
for _ in range(20): ops.matmul(cl, a, bt, c, algo=N.GEMM_ALGO_LP_256)
cl.sync(); dt=(time.perf_counter()-t0)/20
tiles=(S//256)**2; rounds=tiles/256
print('  wall %.4f ms per GEMM; rounds %.2f -> implied clock %.3f GHz (total cycles per tile x rounds / wall)' % (dt*1e3, rounds, t[:, :, 11].mean()*rounds/dt/1e9))
