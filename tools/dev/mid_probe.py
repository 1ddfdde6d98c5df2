"""dev: the 128x128 kernel on mid-size shapes: per-K-tile cost and fixed cost (GPU box)."""
import ctypes as C, sys
sys.path.insert(0, ".")
import bench
from cubecl_amd import ElemType, Mi355Runtime, TensorHandle
from cubecl_amd import _native as N
client = Mi355Runtime.client(); lib, ctx = client.lib, client.ctx
ev = bench.Events(client)
for (m, n, k, algos) in ((2048, 2048, 512, (3, 5)), (2048, 2048, 1024, (3, 5)), (2048, 2048, 2048, (3, 5)), (2048, 2048, 4096, (3, 5)), (2048, 2048, 8192, (3, 5)),
                         (4096, 2048, 2048, (3, 5)), (4096, 4096, 2048, (3, 5, 6)), (1024, 4096, 4096, (3, 5)), (3072, 3072, 3072, (3, 5)), (2560, 2560, 2560, (3, 5))):
    a = TensorHandle.uniform(client, (m, k), ElemType.BF16, 1, 1, -1.0, 1.0); b = TensorHandle.uniform(client, (n, k), ElemType.BF16, 1, 2, -1.0, 1.0)
    c = client.empty(m * n * 2)
    line = []
    for algo in algos:
        d = bench.gemm_desc(N, m, n, k, N.DTYPE_BF16, N.DTYPE_BF16, trans_b=1, algo=algo)
        if lib.mi355_gemm(ctx, None, C.byref(d), a.device_ptr(), b.device_ptr(), c.device_ptr()) != 0:
            line.append(f"algo{algo} --"); continue
        best = min(bench.time_op(client, ev, lambda: lib.mi355_gemm(ctx, None, C.byref(d), a.device_ptr(), b.device_ptr(), c.device_ptr()), 20, warmup=3) for _ in range(5))
        line.append(f"algo{algo} {best * 1e3:7.1f} us {2.0 * m * n * k / best / 1e9:7.1f} TF")
    t128 = ((m + 127) // 128) * ((n + 127) // 128)
    print(f"{m}x{n}x{k} ({t128} tiles of 128^2): " + "   ".join(line), flush=True)
