"""dev: decode-like skinny GEMMs (tokens x out_features x in_features), lp128 split-K vs stream64 vs AUTO (GPU box)."""
import ctypes as C, sys
sys.path.insert(0, ".")
import bench
from cubecl_amd import ElemType, Mi355Runtime, TensorHandle
from cubecl_amd import _native as N
client = Mi355Runtime.client(); lib, ctx = client.lib, client.ctx
ev = bench.Events(client)
NAMES = {3: "lp128", 9: "stream64", 8: "skinny", 5: "w4", 6: "p", 7: "q"}
for (m, n, k) in ((16, 28672, 8192), (64, 28672, 8192), (32, 14336, 4096), (64, 14336, 4096), (16, 4096, 14336), (64, 8192, 28672), (8, 6144, 4096), (64, 6144, 4096), (128, 8192, 8192), (96, 8192, 8192), (128, 28672, 8192), (128, 14336, 4096), (96, 4096, 4096), (8192, 128, 8192), (128, 2048, 2048), (16, 32000, 4096), (64, 128256, 4096)):
    a = TensorHandle.uniform(client, (m, k), ElemType.BF16, 1, 1, -1.0, 1.0); b = TensorHandle.uniform(client, (n, k), ElemType.BF16, 1, 2, -1.0, 1.0)
    c = client.empty(m * n * 2)
    line = []
    for algo in (3, 9, 0):
        d = bench.gemm_desc(N, m, n, k, N.DTYPE_BF16, N.DTYPE_BF16, trans_b=1, algo=algo)
        if lib.mi355_gemm(ctx, None, C.byref(d), a.device_ptr(), b.device_ptr(), c.device_ptr()) != 0:
            line.append(f"{NAMES.get(algo, 'auto')} --"); continue
        best = min(bench.time_op(client, ev, lambda: lib.mi355_gemm(ctx, None, C.byref(d), a.device_ptr(), b.device_ptr(), c.device_ptr()), 20, warmup=3) for _ in range(4))
        sel = C.c_int32(); lib.mi355_gemm_select(ctx, C.byref(d), C.byref(sel))
        nm = NAMES.get(algo, "auto->" + NAMES.get(sel.value, str(sel.value)))
        line.append(f"{nm} {best * 1e3:6.1f} us {2.0 * n * k / best / 1e6:5.0f} GB/s")
    print(f"{m}x{n}x{k}: " + "   ".join(line), flush=True)
