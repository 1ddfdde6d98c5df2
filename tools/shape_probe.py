#!/usr/bin/env python3
"""Times mi355_gemm (AUTO) for a list of bf16 shapes on COLD operands (launches rotate through operand sets that together exceed the
256 MiB Infinity Cache, as bench.py's gemm_bf16_shapes does): median of 5 runs of 20 back-to-back launches.
usage: [MI355CUBE_LIB=...] python tools/shape_probe.py m,n,k[,batch[,nn]] ..."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cubecl_amd import ElemType, Mi355Runtime, TensorHandle
from cubecl_amd import _native as N
cl = Mi355Runtime.client(); lib, ctx = cl.lib, cl.ctx
ev = bench.Events(cl)
for spec in sys.argv[1:]:
    parts = spec.split(",")
    m, n, k = (int(x) for x in parts[:3]); batch = int(parts[3]) if len(parts) > 3 else 1; tb = 0 if (len(parts) > 4 and parts[4] == "nn") else 1
    fp = 2 * (m * k + n * k + m * n) * batch
    nsets = max(1, min(8, -(-(768 << 20) // fp)))
    sets = [(TensorHandle.uniform(cl, (batch, m, k), ElemType.BF16, 1, 700 + 2 * i, -1.0, 1.0), TensorHandle.uniform(cl, (batch, n, k), ElemType.BF16, 1, 701 + 2 * i, -1.0, 1.0),
             cl.empty(batch * m * n * 2)) for i in range(nsets)]
    d = bench.gemm_desc(N, m, n, k, N.DTYPE_BF16, N.DTYPE_BF16, trans_b=tb, batch=batch)
    alg = C.c_int32(); lib.mi355_gemm_select(ctx, C.byref(d), C.byref(alg))
    turn = [0]
    def call():
        a, b, c = sets[turn[0] % nsets]; turn[0] += 1
        cl._s.check(lib.mi355_gemm(ctx, None, C.byref(d), a.device_ptr(), b.device_ptr(), c.device_ptr()))
    iters = 20 if fp < (2 << 30) else 5
    runs = sorted(bench.time_op(cl, ev, call, iters, warmup=3) for _ in range(5))
    print(f"{spec:>24} algo {alg.value:2d}  {runs[2]*1e3:9.1f} us  (min {runs[0]*1e3:9.1f})  {2.0*m*n*k*batch/runs[2]/1e9:8.1f} TFLOP/s", flush=True)
    del sets; cl.memory_cleanup()
