"""dev: the skinny / output-bound shapes north_star names, each candidate kernel timed (GPU box)."""
import ctypes as C, sys
sys.path.insert(0, ".")
import bench
from cubecl_amd import ElemType, Mi355Runtime, TensorHandle
from cubecl_amd import _native as N
client = Mi355Runtime.client(); lib, ctx = client.lib, client.ctx
ev = bench.Events(client)
NAMES = {0: "auto", 3: "lp128", 5: "w4", 6: "p", 7: "q", 8: "skinny", 4: "lp256", 9: "stream64"}
for (m, n, k, algos) in ((1, 8192, 8192, (8, 3)), (4, 8192, 8192, (8, 3, 9)), (16, 8192, 8192, (8, 3, 9)), (32, 8192, 8192, (3, 9)), (8192, 16, 8192, (8, 3, 9)), (1, 16384, 16384, (8, 3)),
                         (64, 8192, 8192, (0, 3, 9)), (8192, 64, 8192, (0, 3, 9)), (64, 16384, 4096, (3, 9)), (48, 4096, 16384, (3, 9)), (64, 2048, 2048, (3, 9)), (8192, 8192, 64, (0, 5, 3, 4)), (8192, 8192, 128, (0, 5, 3, 4)),
                         (2048, 2048, 2048, (0, 3, 5))):
    a = TensorHandle.uniform(client, (m, k), ElemType.BF16, 1, 1, -1.0, 1.0); b = TensorHandle.uniform(client, (n, k), ElemType.BF16, 1, 2, -1.0, 1.0)
    c = client.empty(m * n * 2)
    line = []
    for algo in algos:
        d = bench.gemm_desc(N, m, n, k, N.DTYPE_BF16, N.DTYPE_BF16, trans_b=1, algo=algo)
        if lib.mi355_gemm(ctx, None, C.byref(d), a.device_ptr(), b.device_ptr(), c.device_ptr()) != 0:
            line.append(f"{NAMES[algo]} unsupported"); continue
        sel = C.c_int32(); lib.mi355_gemm_select(ctx, C.byref(d), C.byref(sel))
        best = min(bench.time_op(client, ev, lambda: lib.mi355_gemm(ctx, None, C.byref(d), a.device_ptr(), b.device_ptr(), c.device_ptr()), 20, warmup=3) for _ in range(5))
        gbs = 2.0 * (m * k + n * k + m * n) / best / 1e6
        line.append(f"{NAMES[algo]}{'->' + NAMES.get(sel.value, str(sel.value)) if algo == 0 else ''} {best * 1e3:7.1f} us {2.0 * m * n * k / best / 1e9:7.1f} TF {gbs:6.0f} GB/s")
    print(f"{m}x{n}x{k}: " + "   ".join(line), flush=True)
