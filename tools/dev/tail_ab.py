"""dev: AUTO (may split the last round) vs the plain 256x256 launch, interleaved, bf16."""
import ctypes as C, sys
sys.path.insert(0, ".")
import bench
from cubecl_amd import ElemType, Mi355Runtime, TensorHandle
from cubecl_amd import _native as N
client = Mi355Runtime.client(); lib, ctx = client.lib, client.ctx
ev = bench.Events(client)
for shape in sys.argv[1:]:
    m, n, k = map(int, shape.split("x"))
    a = TensorHandle.uniform(client, (m, k), ElemType.BF16, 1, 1, -1.0, 1.0); b = TensorHandle.uniform(client, (n, k), ElemType.BF16, 1, 2, -1.0, 1.0)
    c = client.empty(m * n * 2)
    calls = {}
    for name, algo in (("auto", 0), ("w4", N.GEMM_ALGO_LP_256W4)):
        d = N.GemmDesc(m=m, n=n, k=k, batch=1, lda=k, ldb=k, ldc=n, dtype_ab=N.DTYPE_BF16, dtype_c=N.DTYPE_BF16, trans_b=1, algo=algo)
        calls[name] = (lambda d=d: client._s.check(lib.mi355_gemm(ctx, None, C.byref(d), a.device_ptr(), b.device_ptr(), c.device_ptr())))
    bench.time_op(client, ev, calls["w4"], 60)
    res = {"auto": [], "w4": []}
    for rep in range(3):
        for name in ("auto", "w4"):
            ms = bench.time_op(client, ev, calls[name], 30)
            res[name].append(2.0 * m * n * k / ms / 1e9)
    au, w4 = sum(res["auto"]) / 3, sum(res["w4"]) / 3
    tiles = ((m + 255) // 256) * ((n + 255) // 256)
    print(f"{shape:>18} tiles {tiles:5d} ({tiles / 256:.2f} rounds)  auto {au:6.0f}  w4 {w4:6.0f}  {100 * (au / w4 - 1):+5.1f} %")
