"""dev: gemm_stream64 (algo 9) on the shapes AUTO sends it, for the library named by MI355CUBE_LIB (GPU box)."""
import ctypes as C, os, sys
sys.path.insert(0, ".")
import bench
from cubecl_amd import ElemType, Mi355Runtime, TensorHandle
from cubecl_amd import _native as N
client = Mi355Runtime.client(); lib, ctx = client.lib, client.ctx
ev = bench.Events(client)
out = []
for (m, n, k) in ((16, 28672, 8192), (32, 14336, 4096), (16, 32000, 4096), (8, 16384, 4096), (32, 24576, 2048), (16, 8192, 8192), (64, 8192, 8192), (64, 14336, 4096)):
    a = TensorHandle.uniform(client, (m, k), ElemType.BF16, 1, 1, -1.0, 1.0); b = TensorHandle.uniform(client, (n, k), ElemType.BF16, 1, 2, -1.0, 1.0)
    c = client.empty(m * n * 2)
    d = bench.gemm_desc(N, m, n, k, N.DTYPE_BF16, N.DTYPE_BF16, trans_b=1, algo=9)
    run = lambda: lib.mi355_gemm(ctx, None, C.byref(d), a.device_ptr(), b.device_ptr(), c.device_ptr())
    assert run() == 0
    best = min(bench.time_op(client, ev, run, 20, warmup=3) for _ in range(5))
    out.append(f"{m}x{n}x{k} {best * 1e3:5.1f}")
print(os.path.basename(os.environ.get("MI355CUBE_LIB", "product")), " | ".join(out), flush=True)
