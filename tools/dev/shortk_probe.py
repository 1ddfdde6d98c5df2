"""dev: output-bound GEMMs (short K) on each candidate kernel (GPU box)."""
import ctypes as C, sys, os
sys.path.insert(0, ".")
import bench
from cubecl_amd import ElemType, Mi355Runtime, TensorHandle
from cubecl_amd import _native as N
client = Mi355Runtime.client(); lib, ctx = client.lib, client.ctx
ev = bench.Events(client)
NAMES = {0: "auto", 3: "lp128", 5: "w4", 6: "p", 7: "q"}
for (m, n, k) in ((8192, 3072, 256), (8192, 3072, 512), (8192, 3072, 1024), (8192, 3072, 2048), (8192, 3072, 4096), (8192, 2560, 512), (8192, 2560, 2048), (4608, 4608, 512), (4608, 4608, 2048), (7168, 4096, 512), (7168, 4096, 1024), (7168, 4096, 4096), (5120, 4352, 512), (5120, 4352, 1536)):
    a = TensorHandle.uniform(client, (m, k), ElemType.BF16, 1, 1, -1.0, 1.0); b = TensorHandle.uniform(client, (n, k), ElemType.BF16, 1, 2, -1.0, 1.0)
    c = client.empty(m * n * 2)
    line = []
    for algo in (3, 5, 6, 7, 0):
        d = bench.gemm_desc(N, m, n, k, N.DTYPE_BF16, N.DTYPE_BF16, trans_b=1, algo=algo)
        if lib.mi355_gemm(ctx, None, C.byref(d), a.device_ptr(), b.device_ptr(), c.device_ptr()) != 0:
            line.append(f"{NAMES[algo]} --"); continue
        best = min(bench.time_op(client, ev, lambda: lib.mi355_gemm(ctx, None, C.byref(d), a.device_ptr(), b.device_ptr(), c.device_ptr()), 20, warmup=3) for _ in range(5))
        sel = C.c_int32(); lib.mi355_gemm_select(ctx, C.byref(d), C.byref(sel))
        line.append(f"{NAMES[algo]}{'->' + NAMES.get(sel.value, str(sel.value)) if algo == 0 else ''} {best * 1e3:6.1f} us")
    print(os.path.basename(os.environ.get("MI355CUBE_LIB", "product")), f"{m}x{n}x{k}: " + "  ".join(line), flush=True)
