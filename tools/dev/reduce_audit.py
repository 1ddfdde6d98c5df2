#!/usr/bin/env python3
"""Dev (GPU box): seeded random (outer, axis length, inner) shapes through mi355_reduce_axis / mi355_argreduce_axis against the HBM roofline -- the reductions have no forced
kernels to compare with, so the yardstick is input bytes / time over 8 TB/s; prints the shapes under 0.35 of it that take more than 10 us (a launch is ~6 us).
usage: tools/dev/reduce_audit.py [seed] [count]"""
import math, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from cubecl_amd import ElemType, Mi355Runtime, TensorHandle, ops
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = random.Random(seed)
cl = Mi355Runtime.client(); ev = bench.Events(cl)
def dim(lo, hi): return max(1, int(round(math.exp(rng.uniform(math.log(lo), math.log(hi))))))
shapes = []
while len(shapes) < count:
    kind = rng.choice(["last", "last", "mid", "mid", "first"])
    if kind == "last": outer, length, inner = dim(1, 1 << 20), dim(2, 1 << 20), 1
    elif kind == "first": outer, length, inner = 1, dim(2, 1 << 16), dim(2, 1 << 20)
    else: outer, length, inner = dim(1, 1 << 14), dim(2, 1 << 14), dim(2, 1 << 14)
    n = outer * length * inner
    if n < (1 << 21) or n > (1 << 28): continue
    shapes.append((outer, length, inner))
slow = 0
for dtype in (ElemType.F32, ElemType.BF16):
    for (outer, length, inner) in shapes:
        n = outer * length * inner
        x = TensorHandle.uniform(cl, (outer, length, inner), dtype, 1, 900, -1.0, 1.0)
        o = TensorHandle.new_contiguous((outer, inner), cl.empty(outer * inner * 4), ElemType.F32)
        oi = TensorHandle.new_contiguous((outer, inner), cl.empty(outer * inner * 4), ElemType.U32)
        line = f"{dtype.name:5s} ({outer:8d},{length:8d},{inner:8d}): "
        flag = ""
        for op in ("sum", "argmax"):
            fn = (lambda: ops.argreduce_axis(cl, x, oi, 1, op)) if op.startswith("arg") else (lambda: ops.reduce_axis(cl, x, o, 1, op))
            us = bench.time_op(cl, ev, fn, 10, warmup=2) * 1e3          # (time_op gives milliseconds)
            frac = n * dtype.size() / (us * 1e-6) / 8.0e12
            line += f"{op} {us:8.1f} us {frac:5.2f} of HBM   "
            if frac < 0.35 and us > 10.0: flag = "  <-- SLOW"
        slow += bool(flag)
        print(line + flag, flush=True)
        del x, o, oi
    cl.memory_cleanup()
print(f"{slow} of {2 * len(shapes)} under 0.35 of the HBM roofline and over 10 us")
