#!/usr/bin/env python3
"""Dev: time mi355_gemm for explicit shapes / leading dimensions.
usage (GPU box): [MI355CUBE_LIB=...] [GEMM_PROBE_LAYOUT=nt|nn|tn] python tools/dev/gemm_probe.py ALGO m,n,k[,lda,ldb] ...
(nn: rhs row-major [K][N]; tn: lhs stored [K][M] and rhs row-major; leading dimensions default to the layout's row length)"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cubecl_amd import Mi355Runtime, TensorHandle, ElemType
from cubecl_amd import _native as N
cl = Mi355Runtime.client(); lib, ctx = cl.lib, cl.ctx
algo = int(sys.argv[1])
ea, eb = C.c_void_p(), C.c_void_p()
lib.mi355_event_create(ctx, C.byref(ea)); lib.mi355_event_create(ctx, C.byref(eb))
for spec in sys.argv[2:]:
    v = [int(x) for x in spec.split(",")]
    lay = os.environ.get("GEMM_PROBE_LAYOUT", "nt")
    ta, tb = int(lay[0] == "t"), int(lay[1] == "t")
    m, n, k = v[:3]; lda = v[3] if len(v) > 3 else (m if ta else k); ldb = v[4] if len(v) > 4 else (k if tb else n)
    a = TensorHandle.uniform(cl, ((k if ta else m) * lda,), ElemType.BF16, 1, 1, -1.0, 1.0)
    b = TensorHandle.uniform(cl, ((n if tb else k) * ldb,), ElemType.BF16, 1, 2, -1.0, 1.0)
    c = cl.empty(m * n * 2)
    d = N.GemmDesc(m=m, n=n, k=k, batch=1, lda=lda, ldb=ldb, ldc=n, stride_a=(k if ta else m) * lda, stride_b=(n if tb else k) * ldb, stride_c=m * n,
                   dtype_ab=N.DTYPE_BF16, dtype_c=N.DTYPE_BF16, trans_a=ta, trans_b=tb, algo=algo)
    pa, pb, pc = C.c_void_p(a.device_ptr()), C.c_void_p(b.device_ptr()), C.c_void_p(c.device_ptr())
    for _ in range(3): cl._s.check(lib.mi355_gemm(ctx, None, C.byref(d), pa, pb, pc))
    cl.sync()
    best = 1e9; tot = 0
    for _ in range(3):
        lib.mi355_event_record(ctx, ea, None)
        for _ in range(10): lib.mi355_gemm(ctx, None, C.byref(d), pa, pb, pc)
        lib.mi355_event_record(ctx, eb, None); lib.mi355_event_sync(ctx, eb)
        ms = C.c_float(); lib.mi355_event_elapsed_ms(ctx, ea, eb, C.byref(ms)); best = min(best, ms.value / 10)
    tiles = (m // 256) * (n // 256); nk = k // 64
    gbs_cu = tiles * nk * 65536 / (best * 1e-3) / 1e9 / 256
    print(f"{lay} algo {algo} {m}x{n}x{k} lda {lda} ldb {ldb}: {best:.4f} ms  {2.0*m*n*k/best/1e9:8.1f} TF   L2->LDS {gbs_cu:6.1f} GB/s/CU", flush=True)
    del a, b, c
    cl.flush()
