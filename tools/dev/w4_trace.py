#!/usr/bin/env python3
"""Dev: where a lp256w4 workgroup's cycles go (needs a -DW4_TRACE variant build).
usage (GPU box): MI355CUBE_LIB=.../libmi355cube_w4trace.so python tools/dev/w4_trace.py m,n,k[,batch] ..."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from cubecl_amd import Mi355Runtime, TensorHandle, ElemType
from cubecl_amd import _native as N
cl = Mi355Runtime.client(); lib, ctx = cl.lib, cl.ctx
for spec in sys.argv[1:]:
    v = [int(x) for x in spec.split(",")]; m, n, k = v[:3]; batch = v[3] if len(v) > 3 else 1
    a = TensorHandle.uniform(cl, (batch * m * k,), ElemType.BF16, 1, 1, -1.0, 1.0); b = TensorHandle.uniform(cl, (batch * n * k,), ElemType.BF16, 1, 2, -1.0, 1.0)
    c = cl.empty(batch * m * n * 2)
    d = N.GemmDesc(m=m, n=n, k=k, batch=batch, lda=k, ldb=k, ldc=n, stride_a=m*k, stride_b=n*k, stride_c=m*n, dtype_ab=N.DTYPE_BF16, dtype_c=N.DTYPE_BF16, trans_a=0, trans_b=1, algo=5)
    for _ in range(60): lib.mi355_gemm(ctx, None, C.byref(d), C.c_void_p(a.device_ptr()), C.c_void_p(b.device_ptr()), C.c_void_p(c.device_ptr()))
    cl.sync()
    buf = np.zeros(4096 * 8, dtype=np.uint64)
    lib.mi355_dev_w4_trace(buf.ctypes.data_as(C.c_void_p))
    tiles = min((m // 256) * (n // 256), 4096)
    t = buf.reshape(4096, 8)[:tiles].astype(np.float64)
    pro, loop, epi = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
    nk = k // 64
    tot = t[:, 3] - t[:, 0]
    print(f"{spec}: tiles {tiles} nk {nk}: prologue {np.median(pro):8.0f}  loop {np.median(loop):9.0f} ({np.median(loop)/nk:7.1f}/K-tile, floor 2048)  "
          f"epilogue+drain {np.median(epi):8.0f}  total {np.median(tot):9.0f} cycles; p10/p90 of total {np.percentile(tot,10):.0f}/{np.percentile(tot,90):.0f}", flush=True)
    rt0, rt1 = t[:, 4], t[:, 5]                       # 100 MHz constant clock, comparable across CUs
    span_us = (rt1.max() - rt0.min()) / 100.0
    busy_us = (rt1 - rt0).sum() / 100.0 / 256         # mean busy time per CU
    order = np.argsort(rt0)
    starts = (rt0[order] - rt0.min()) / 100.0
    print(f"   wall (100 MHz clock): first start -> last end {span_us:.1f} us; mean CU busy {busy_us:.1f} us ({busy_us/span_us:.1%}); "
          f"block durations us p10/50/90 {np.percentile(rt1-rt0,10)/100:.1f}/{np.percentile(rt1-rt0,50)/100:.1f}/{np.percentile(rt1-rt0,90)/100:.1f}; "
          f"start of block #255/#256/#511/#512/#767/#768: " + "/".join(f"{starts[i]:.1f}" for i in (255, 256, 511, 512, 767, 768) if i < len(starts)))
    del a, b, c; cl.flush()
