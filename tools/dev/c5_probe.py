"""dev: batched 2048^3 bf16 (config C5 shard) and 4096^3 on every kernel that takes them."""
import ctypes as C, json, sys
sys.path.insert(0, ".")
import bench
from cubecl_amd import ElemType, Mi355Runtime, TensorHandle
from cubecl_amd import _native as N
client = Mi355Runtime.client(); lib, ctx = client.lib, client.ctx
ev = bench.Events(client)
out = {}
for (M, batch) in ((2048, 64), (4096, 1), (1024, 256)):
    a = TensorHandle.uniform(client, (batch, M, M), ElemType.BF16, 1, 500, -1.0, 1.0)
    b = TensorHandle.uniform(client, (batch, M, M), ElemType.BF16, 1, 600, -1.0, 1.0)
    c = client.empty(batch * M * M * 2)
    for name, algo in (("auto", 0), ("lp128", N.GEMM_ALGO_LP_128), ("lp256", N.GEMM_ALGO_LP_256), ("w4", N.GEMM_ALGO_LP_256W4), ("p", N.GEMM_ALGO_LP_256P)):
        d = bench.gemm_desc(N, M, M, M, N.DTYPE_BF16, N.DTYPE_BF16, trans_b=1, batch=batch, algo=algo)
        call = lambda: client._s.check(lib.mi355_gemm(ctx, None, C.byref(d), a.device_ptr(), b.device_ptr(), c.device_ptr()))
        try:
            bench.time_op(client, ev, call, 20)
            ms = bench.time_op(client, ev, call, 20)
            out[f"{M}x{batch}_{name}"] = round(2.0 * M ** 3 * batch / ms / 1e9, 1)
        except Exception as e:
            out[f"{M}x{batch}_{name}"] = str(e)[:60]
print(json.dumps(out, indent=1))
