#!/usr/bin/env python3
"""Dev: secondary entry points against the HBM roofline (round 4 looked for cliffs like the one the non-last-axis reductions had)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from cubecl_amd import ElemType, Mi355Runtime, TensorHandle, ops
from cubecl_amd import _native as N
cl = Mi355Runtime.client(); lib, ctx = cl.lib, cl.ctx
ev = bench.Events(cl)
def t(name, fn, nbytes):
    med, best = bench.samples_op(cl, ev, fn)
    print(f"{name:52s} {med * 1e3:9.1f} us  {nbytes / med / 1e6:8.0f} GB/s", flush=True)
# last axis with short rows
for shape in ((1 << 20, 64), (1 << 22, 16), (1 << 24, 4), (1 << 18, 300), (1 << 16, 1000)):
    n = shape[0] * shape[1]
    x = TensorHandle.uniform(cl, shape, ElemType.F32, 1, 900, -1.0, 1.0)
    o = TensorHandle.new_contiguous((shape[0],), cl.empty(shape[0] * 4), ElemType.F32)
    oi = TensorHandle.new_contiguous((shape[0],), cl.empty(shape[0] * 4), ElemType.U32)
    t(f"last axis sum    {shape}", lambda: ops.reduce_axis(cl, x, o, 1, "sum"), n * 4)
    t(f"last axis argmax {shape}", lambda: ops.argreduce_axis(cl, x, oi, 1, "argmax"), n * 4)
    del x, o, oi
n = 1 << 26
x = TensorHandle.uniform(cl, (n,), ElemType.F32, 1, 901, -1.0, 1.0)
o = TensorHandle.new_contiguous((n,), cl.empty(n * 4), ElemType.F32)
for op, name in ((N.REDUCE_SUM, "plane_sum"), (N.PLANE_INCLUSIVE_SUM, "plane_inclusive_sum"), (N.PLANE_EXCLUSIVE_PROD, "plane_exclusive_prod")):
    t(f"{name} 256 MiB (read + write)", lambda: ops.plane_reduce(cl, x, o, op, active=64), 2 * n * 4)
t("plane_op shuffle_xor 256 MiB (read + write)", lambda: ops.plane_op(cl, x, o, N.PLANE_SHUFFLE_XOR, plane=64, arg=1), 2 * n * 4)
ob = cl.empty(n * 2)
t("cast f32 -> bf16 256 MiB in", lambda: cl._s.check(lib.mi355_cast(ctx, None, C.c_void_p(x.device_ptr()), N.DTYPE_F32, C.c_void_p(ob.device_ptr()), N.DTYPE_BF16, n)), n * 6)
t("fill_uniform f32 256 MiB", lambda: cl._s.check(lib.mi355_fill_uniform(ctx, None, C.c_void_p(o.device_ptr()), N.DTYPE_F32, n, 1, 5, -1.0, 1.0)), n * 4)
n2 = 1 << 29
xb = TensorHandle.uniform(cl, (n2,), ElemType.BF16, 1, 902, -1.0, 1.0)
s1 = TensorHandle.new_contiguous((1,), cl.empty(8), ElemType.F32); i1 = TensorHandle.new_contiguous((1,), cl.empty(8), ElemType.U64)
t("array-wide sum bf16 1 GiB", lambda: ops.reduce(cl, xb, s1, "sum"), n2 * 2)
t("array-wide max bf16 1 GiB", lambda: ops.reduce(cl, xb, s1, "max"), n2 * 2)
t("array-wide argmin bf16 1 GiB", lambda: ops.argreduce(cl, xb, i1, None, "argmin"), n2 * 2)
# f32 GEMM below the 256^2 kernel's range
for (m, nn, k) in ((2048, 2048, 2048), (1024, 1024, 1024), (4096, 1024, 4096), (512, 8192, 512), (512, 512, 4096), (1536, 1536, 1536), (1024, 2048, 2048)):
    a = TensorHandle.uniform(cl, (m, k), ElemType.F32, 1, 903, -1.0, 1.0); b = TensorHandle.uniform(cl, (nn, k), ElemType.F32, 1, 904, -1.0, 1.0)
    c = cl.empty(m * nn * 4)
    d = bench.gemm_desc(N, m, nn, k, N.DTYPE_F32, N.DTYPE_F32, trans_b=1)
    alg = C.c_int32(); lib.mi355_gemm_select(ctx, C.byref(d), C.byref(alg))
    ms = bench.time_op(cl, ev, lambda: cl._s.check(lib.mi355_gemm(ctx, None, C.byref(d), a.device_ptr(), b.device_ptr(), c.device_ptr())), 20)
    print(f"f32 GEMM {m}x{nn}x{k} algo {alg.value}: {ms * 1e3:8.1f} us  {2.0 * m * nn * k / ms / 1e9:7.1f} TFLOP/s ({2.0 * m * nn * k / ms / 1e9 / 157.3:.3f} of 157.3)", flush=True)
