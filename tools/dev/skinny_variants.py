"""dev: skinny kernel variants (GPU box): MI355CUBE_LIB=... python tools/dev/skinny_variants.py"""
import ctypes as C, sys, os
sys.path.insert(0, ".")
import bench
from cubecl_amd import ElemType, Mi355Runtime, TensorHandle, ops
from cubecl_amd import _native as N
client = Mi355Runtime.client(); lib, ctx = client.lib, client.ctx
ev = bench.Events(client)
out = []
for (m, n, k) in ((1, 8192, 8192), (2, 8192, 8192), (4, 8192, 8192), (1, 16384, 16384)):
    a = TensorHandle.uniform(client, (m, k), ElemType.BF16, 1, 1, -1.0, 1.0); b = TensorHandle.uniform(client, (n, k), ElemType.BF16, 1, 2, -1.0, 1.0)
    c = client.empty(m * n * 2)
    d = bench.gemm_desc(N, m, n, k, N.DTYPE_BF16, N.DTYPE_BF16, trans_b=1, algo=8)
    best = min(bench.time_op(client, ev, lambda: lib.mi355_gemm(ctx, None, C.byref(d), a.device_ptr(), b.device_ptr(), c.device_ptr()), 20, warmup=3) for _ in range(5))
    out.append(f"{m}x{n}: {best * 1e3:6.1f} us {2.0 * n * k / best / 1e6:5.0f} GB/s")
if len(sys.argv) > 1:   # the read floor at the same size: the bf16 sum over 128 MiB / 512 MiB
    for elems in (8192 * 8192, 16384 * 16384):
        x = TensorHandle.uniform(client, (elems,), ElemType.BF16, 1, 3, -1.0, 1.0)
        o = TensorHandle.new_contiguous((1,), client.empty(4), ElemType.F32)
        best = min(bench.time_op(client, ev, lambda: ops.reduce_sum(client, x, o), 20, warmup=3) for _ in range(5))
        out.append(f"sum {elems >> 20}Mi bf16: {best * 1e3:6.1f} us {2.0 * elems / best / 1e6:5.0f} GB/s")
print(os.path.basename(os.environ.get("MI355CUBE_LIB", "product")), " | ".join(out), flush=True)
