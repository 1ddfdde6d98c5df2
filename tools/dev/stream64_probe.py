"""dev: gemm_stream64 against the split-K 128x128 path over a grid of skinny shapes (GPU box)."""
import ctypes as C, sys
sys.path.insert(0, ".")
import bench
from cubecl_amd import ElemType, Mi355Runtime, TensorHandle
from cubecl_amd import _native as N
client = Mi355Runtime.client(); lib, ctx = client.lib, client.ctx
ev = bench.Events(client)
shapes = []
for m in (4, 16, 32, 64):
    for (n, k) in ((1024, 1024), (2048, 2048), (4096, 2048), (2048, 4096), (4096, 4096), (8192, 2048), (2048, 8192), (8192, 4096), (4096, 8192), (16384, 2048), (8192, 8192)):
        shapes.append((m, n, k))
shapes += [(64, 16384, 8192), (64, 32768, 4096), (16, 65536, 1024), (64, 4096, 16384), (32, 6144, 8192), (64, 12288, 4096), (2048, 64, 2048), (4096, 32, 4096), (8192, 16, 2048), (64, 512, 512), (64, 1024, 4096), (64, 256, 8192)]
for (m, n, k) in shapes:
    a = TensorHandle.uniform(client, (m, k), ElemType.BF16, 1, 1, -1.0, 1.0); b = TensorHandle.uniform(client, (n, k), ElemType.BF16, 1, 2, -1.0, 1.0)
    c = client.empty(m * n * 2)
    res = {}
    for algo in (3, 9):
        d = bench.gemm_desc(N, m, n, k, N.DTYPE_BF16, N.DTYPE_BF16, trans_b=1, algo=algo)
        if lib.mi355_gemm(ctx, None, C.byref(d), a.device_ptr(), b.device_ptr(), c.device_ptr()) != 0:
            res[algo] = float("nan"); continue
        res[algo] = min(bench.time_op(client, ev, lambda: lib.mi355_gemm(ctx, None, C.byref(d), a.device_ptr(), b.device_ptr(), c.device_ptr()), 20, warmup=3) for _ in range(4)) * 1e3
    mb = 2.0 * max(m, n) * k / 1e6
    print(f"{m}x{n}x{k} ({mb:6.1f} MB streamed): lp128 {res[3]:6.1f} us  stream64 {res[9]:6.1f} us  {'<-- stream64' if res[9] < res[3] * 0.97 else ''}", flush=True)
