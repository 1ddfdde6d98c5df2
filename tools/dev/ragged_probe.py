#!/usr/bin/env python3
"""Dev: bf16 GEMM on shapes that are not multiples of any tile (AUTO), TFLOP/s against the nearest round shape."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from cubecl_amd import ElemType, Mi355Runtime, TensorHandle
from cubecl_amd import _native as N
cl = Mi355Runtime.client(); lib, ctx = cl.lib, cl.ctx
ev = bench.Events(cl)
SHAPES = [(4096, 4096, 4096, 1), (4000, 4000, 4000, 1), (4096, 4096, 4100, 1), (4100, 4100, 4096, 1), (8191, 8191, 8192, 1), (8192, 8192, 8200, 1), (1000, 1000, 1000, 1), (3000, 3000, 3000, 1),
          (2048, 2048, 2080, 1), (512, 512, 512, 64), (256, 256, 256, 256), (128, 128, 128, 1024), (1024, 1024, 1024, 16), (384, 384, 384, 128), (64, 64, 64, 4096)]
for (m, n, k, batch) in SHAPES:
    for tb in (1, 0):
        a = TensorHandle.uniform(cl, (batch, m, k), ElemType.BF16, 1, 1, -1.0, 1.0)
        b = TensorHandle.uniform(cl, (batch, n, k), ElemType.BF16, 1, 2, -1.0, 1.0)
        c = cl.empty(batch * m * n * 2)
        d = bench.gemm_desc(N, m, n, k, N.DTYPE_BF16, N.DTYPE_BF16, trans_b=tb, batch=batch)
        alg = C.c_int32(); lib.mi355_gemm_select(ctx, C.byref(d), C.byref(alg))
        ra, rb = C.c_int32(), C.c_int32(); lib.mi355_gemm_relayout_plan(C.byref(d), C.byref(ra), C.byref(rb))
        rc = lib.mi355_gemm(ctx, None, C.byref(d), a.device_ptr(), b.device_ptr(), c.device_ptr())
        if rc != 0:
            print(f"{m}x{n}x{k} x{batch} {'NT' if tb else 'NN'}: error {rc}"); continue
        ms = sorted(bench.time_op(cl, ev, lambda: lib.mi355_gemm(ctx, None, C.byref(d), a.device_ptr(), b.device_ptr(), c.device_ptr()), 10, warmup=2) for _ in range(3))[1]
        print(f"{m:5d}x{n:5d}x{k:5d} x{batch:<5d} {'NT' if tb else 'NN'} algo {alg.value:2d} relaid {ra.value}{rb.value}: {ms * 1e3:9.1f} us {2.0 * m * n * k * batch / ms / 1e9:8.1f} TFLOP/s", flush=True)
        del a, b, c
    cl.memory_cleanup()
