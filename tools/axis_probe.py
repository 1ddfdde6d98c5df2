#!/usr/bin/env python3
"""Times the reductions over one axis (mi355_reduce_axis / mi355_argreduce_axis) and the array-wide operations of round 4 against the
HBM roofline: algorithmic bytes = the input read once; median of 15 samples (5 warm-ups), sync around every sample.
usage: python tools/axis_probe.py [quick]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cubecl_amd import ElemType, Mi355Runtime, TensorHandle, ops
from cubecl_amd import _native as N
cl = Mi355Runtime.client(); lib, ctx = cl.lib, cl.ctx
ev = bench.Events(cl)
CASES = [((64, 256, 1024), 2), ((64, 256, 1024), 1), ((64, 256, 1024), 0), ((64, 64, 4096), 1), ((512, 8192), 0), ((8192, 8192), 1), ((8192, 8192), 0),
         ((16, 4096, 4096), 1), ((4096, 16, 4096), 1), ((4, 65536, 1024), 1), ((1 << 14, 1 << 14), 0), ((256, 1 << 20), 1),
         ((1 << 20, 16, 4), 1), ((1 << 16, 8, 64), 1), ((1 << 20, 3, 8), 1), ((1 << 18, 128, 2), 1)]       # short axis under a narrow inner
for dtype in (ElemType.F32, ElemType.BF16):
    for shape, axis in CASES:
        n = 1
        for d in shape:
            n *= d
        x = TensorHandle.uniform(cl, shape, dtype, 1, 900, -1.0, 1.0)
        out_shape = tuple(d for i, d in enumerate(shape) if i != axis)
        m = 1
        for d in out_shape:
            m *= d
        o = TensorHandle.new_contiguous(out_shape, cl.empty(m * 4), ElemType.F32)
        oi = TensorHandle.new_contiguous(out_shape, cl.empty(m * 4), ElemType.U32)
        line = f"{dtype.name:5s} {str(shape):>22s} axis {axis}: "
        for op in ("sum", "max", "argmax"):
            fn = (lambda: ops.argreduce_axis(cl, x, oi, axis, op)) if op.startswith("arg") else (lambda: ops.reduce_axis(cl, x, o, axis, op))
            med, best = bench.samples_op(cl, ev, fn)
            line += f"{op} {med * 1e3:8.1f} us {n * dtype.size() / med / 1e6:7.0f} GB/s   "
        print(line, flush=True)
        del x, o, oi
    cl.memory_cleanup()
n = 1 << 28
x = TensorHandle.uniform(cl, (n,), ElemType.F32, 1, 300, 0.0, 1.0)
out = TensorHandle.new_contiguous((1,), cl.empty(8), ElemType.F32)
idx = TensorHandle.new_contiguous((1,), cl.empty(8), ElemType.U64)
for op in ("sum", "mean", "max", "min", "prod", "argmax", "argmin"):
    fn = (lambda: ops.argreduce(cl, x, idx, None, op)) if op.startswith("arg") else (lambda: ops.reduce(cl, x, out, op))
    med, best = bench.samples_op(cl, ev, fn)
    b2b = bench.time_op(cl, ev, fn, 20, warmup=2)
    print(f"array-wide 1 GiB f32 {op:7s}: median {med * 1e3:7.1f} us {n * 4 / med / 1e6:7.0f} GB/s   back to back {b2b * 1e3:7.1f} us {n * 4 / b2b / 1e6:7.0f} GB/s", flush=True)
