#!/usr/bin/env python3
"""Launches BASELINE config C5 as bench.py times it at N = 1 -- batch 512 x 2048^3 bf16 -> bf16 C, AUTO (gemm_lp256q.hip) -- a few
times; the command the C5 PMC passes of tools/pmc_all.sh profile.  Prints the rate and the shader clock the launches ran at
(mi355_probe_clock around them: s_memtime against the 100 MHz reference, per CU).
usage: python tools/c5_probe.py [launches] [nn] [batch] [algo]     (algo: MI355_GEMM_ALGO_* to force, default AUTO)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from cubecl_amd import ElemType, Mi355Runtime, TensorHandle
from cubecl_amd import _native as N
launches = int(sys.argv[1]) if len(sys.argv) > 1 else 6
tb = 0 if (len(sys.argv) > 2 and sys.argv[2] == "nn") else 1
B = int(sys.argv[3]) if len(sys.argv) > 3 else 512
M = int(os.environ.get("PROBE_M", 2048))
cl = Mi355Runtime.client(); lib, ctx = cl.lib, cl.ctx
ev = bench.Events(cl)
a = TensorHandle.uniform(cl, (B, M, M), ElemType.BF16, bench.SEED, 500, -1.0, 1.0)
b = TensorHandle.uniform(cl, (B, M, M), ElemType.BF16, bench.SEED, 600, -1.0, 1.0)
c = cl.empty(B * M * M * 2)
FORCE = int(sys.argv[4]) if len(sys.argv) > 4 else 0
d = bench.gemm_desc(N, M, M, M, N.DTYPE_BF16, N.DTYPE_BF16, trans_b=tb, batch=B, algo=FORCE)
alg = C.c_int32(FORCE)
if not FORCE:
    lib.mi355_gemm_select(ctx, C.byref(d), C.byref(alg))
call = lambda: cl._s.check(lib.mi355_gemm(ctx, None, C.byref(d), a.device_ptr(), b.device_ptr(), c.device_ptr()))
clk = cl.empty(2 * 8192)
lib.mi355_memset(ctx, None, C.c_void_p(clk.device_ptr()), 0, 2 * 8192)
p0, p1 = C.c_void_p(clk.device_ptr()), C.c_void_p(clk.device_ptr() + 8192)
for _ in range(3):
    call()
lib.mi355_probe_clock(ctx, None, p0)
cl.sync()
lib.mi355_probe_clock(ctx, None, p0)
ms = bench.time_op(cl, ev, call, launches, warmup=0)
lib.mi355_probe_clock(ctx, None, p1)
tk = np.frombuffer(cl.read_one(clk), dtype=np.uint64).reshape(2, 512, 2).astype(np.float64)
ok = (tk[0, :, 1] > 0) & (tk[1, :, 1] > tk[0, :, 1]) & (tk[1, :, 0] > tk[0, :, 0])
ghz = float(np.median((tk[1, ok, 0] - tk[0, ok, 0]) / (tk[1, ok, 1] - tk[0, ok, 1]) * 0.1)) if ok.any() else float("nan")
tf = 2.0 * M ** 3 * B / ms / 1e9
print(f"C5 {B} x {M}^3 bf16 {'NT' if tb else 'NN'} algo {alg.value}: {ms:.3f} ms/launch  {tf:.1f} TFLOP/s  frac {tf / 2500:.4f}  "
      f"shader clock {ghz:.3f} GHz  frac_at_clock {tf / (2500 * ghz / 2.4):.4f}", flush=True)
