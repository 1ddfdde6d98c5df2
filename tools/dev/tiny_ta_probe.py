#!/usr/bin/env python3
"""Dev (GPU box): AUTO on tiny products whose layout no MFMA kernel takes as it is (lhs stored [K][M], rows not a multiple of 8) -- the scalar kernel against the re-layout path
(plan_relayout's tiny bound, late round 6).  usage: tools/dev/tiny_ta_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ab_algos, bench
from cubecl_amd import Mi355Runtime
cl = Mi355Runtime.client(); ev = bench.Events(cl)
shapes = [(1, 2360, 512), (304, 2, 1024), (288, 5, 1024), (10, 304, 256), (5, 488, 256), (11, 792, 128), (14, 328, 64), (2, 496, 128), (33, 65, 2000), (100, 20, 1024), (7, 7, 8192)]
for ta in (True, False):
    res = ab_algos.measure(cl, ev, shapes, ["auto", "generic"], rounds=3, iters=10, nn=True, ta=ta)
    print("lhs [K][M] x row-major rhs" if ta else "lhs [M][K] x row-major rhs")
    for (m, n, k), r in res.items():
        print(f"  {m:5d}x{n:5d}x{k:5d}: AUTO -> {r['auto']:9s} {r['us']['auto']:7.1f} us   scalar kernel forced {r['us']['generic']:7.1f} us")
