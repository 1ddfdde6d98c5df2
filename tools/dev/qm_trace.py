#!/usr/bin/env python3
"""Dev: where a workgroup of the persistent dripped-store kernel (gemm_lp256qm.hip) spends its cycles on config C5 -- K loop against
tile boundary, per tile (needs a -DQM_TRACE variant build: tools/dev/build_variants.sh qmtrace "-DQM_TRACE" gemm_lp256qm.hip).
Resolution: a stamp (s_memtime + its wait + a store) costs ~1 000 cycles itself -- the "gap" column is two stamps and little else,
and every K-loop figure carries one (profiles/r04_c5_counters.md).
usage (GPU box): MI355CUBE_LIB=$PWD/cubecl_amd/csrc/variants/libmi355cube_qmtrace.so python tools/dev/qm_trace.py [batch]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from cubecl_amd import ElemType, Mi355Runtime, TensorHandle
from cubecl_amd import _native as N
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
M = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
NT = min(8, B * (M // 256) ** 2 // 256)
cl = Mi355Runtime.client(); lib, ctx = cl.lib, cl.ctx
a = TensorHandle.uniform(cl, (B, M, M), ElemType.BF16, 1, 1, -1.0, 1.0); b = TensorHandle.uniform(cl, (B, M, M), ElemType.BF16, 1, 2, -1.0, 1.0)
c = cl.empty(B * M * M * 2)
for tb in (1,):
    d = bench.gemm_desc(N, M, M, M, N.DTYPE_BF16, N.DTYPE_BF16, trans_b=tb, batch=B, algo=N.GEMM_ALGO_LP_256QM)
    for _ in range(4):
        cl._s.check(lib.mi355_gemm(ctx, None, C.byref(d), a.device_ptr(), b.device_ptr(), c.device_ptr()))
    cl.sync()
    buf = np.zeros(256 * 64, dtype=np.uint64)
    lib.mi355_dev_qm_trace(buf.ctypes.data_as(C.c_void_p))
    w = buf.reshape(256, 64)[:, 32:52].astype(np.float64).reshape(256, 5, 4)
    t = buf.reshape(256, 64)[:, :24].astype(np.float64).reshape(256, 8, 3)
    loop = t[:, :, 1] - t[:, :, 0]            # K loop of tile i (32 K-tiles)
    bound = t[:, :, 2] - t[:, :, 1]           # its boundary (block rows 6-7 through LDS + 8 stores)
    gap = t[:, 1:, 0] - t[:, :-1, 2]          # boundary end -> next K loop entered (locate(), scalar set-up)
    nk = M // 64
    print(f"{'NT' if tb else 'NN'} batch {B}: per tile (median over 256 workgroups), shader cycles")
    for i in range(NT):
        print(f"  tile {i}: K loop {np.median(loop[:, i]):8.0f} = {np.median(loop[:, i]) / nk:6.0f} per K-tile (floor 2048)   boundary {np.median(bound[:, i]):6.0f}"
              + (f"   gap {np.median(gap[:, i - 1]):5.0f}" if i else ""))
    tot = np.median(t[:, NT - 1, 2] - t[:, 0, 0])
    print(f"  {NT} tiles: {tot:.0f} cycles; K loops {np.median(loop[:, :NT].sum(axis=1)) / tot:.4f}, boundaries {np.median(bound[:, :NT].sum(axis=1)) / tot:.4f}, "
          f"MFMA floor share {NT * nk * 2048 / tot:.4f}", flush=True)
    # cycles per K-tile inside the hand-over (stamps included: ~2 s_memtime round trips), per wave, median over workgroups
    cnt = np.maximum(w[:, 4, :], 1)
    print("  hand-over per K-tile, first tile:  vmcnt wait " + " ".join(f"{np.median(w[:, 0, i]) / nk:6.0f}" for i in range(4))
          + "   lgkm + barrier " + " ".join(f"{np.median(w[:, 1, i]) / nk:6.0f}" for i in range(4)))
    print("  hand-over per K-tile, later tiles: vmcnt wait " + " ".join(f"{np.median(w[:, 2, i] / cnt[:, i]):6.0f}" for i in range(4))
          + "   lgkm + barrier " + " ".join(f"{np.median(w[:, 3, i] / cnt[:, i]):6.0f}" for i in range(4)), flush=True)
