"""dev: persistent (lp256p) vs one-tile-per-workgroup (lp256w4), interleaved, bf16."""
import ctypes as C, sys
sys.path.insert(0, ".")
import bench
from cubecl_amd import ElemType, Mi355Runtime, TensorHandle
from cubecl_amd import _native as N
client = Mi355Runtime.client(); lib, ctx = client.lib, client.ctx
ev = bench.Events(client)
for spec in sys.argv[1:]:
    parts = list(map(int, spec.split("x")))
    m, n, k = parts[:3]; batch = parts[3] if len(parts) > 3 else 1
    a = TensorHandle.uniform(client, (batch, m, k), ElemType.BF16, 1, 1, -1.0, 1.0); b = TensorHandle.uniform(client, (batch, n, k), ElemType.BF16, 1, 2, -1.0, 1.0)
    c = client.empty(batch * m * n * 2)
    calls = {}
    for name, algo in (("p", N.GEMM_ALGO_LP_256P), ("w4", N.GEMM_ALGO_LP_256W4)):
        d = bench.gemm_desc(N, m, n, k, N.DTYPE_BF16, N.DTYPE_BF16, batch=batch, algo=algo)
        calls[name] = (lambda d=d: client._s.check(lib.mi355_gemm(ctx, None, C.byref(d), a.device_ptr(), b.device_ptr(), c.device_ptr())))
    bench.time_op(client, ev, calls["w4"], 40)
    res = {"p": [], "w4": []}
    for rep in range(3):
        for name in ("p", "w4"):
            ms = bench.time_op(client, ev, calls[name], 20)
            res[name].append(2.0 * m * n * k * batch / ms / 1e9)
    p, w = sum(res["p"]) / 3, sum(res["w4"]) / 3
    print(f"{spec:>20}  p {p:6.0f}  w4 {w:6.0f}  {100 * (p / w - 1):+5.1f} %")
