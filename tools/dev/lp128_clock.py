"""dev: sustained shader clock during back-to-back launches of the 128x128 kernel (product and ablation builds via
MI355CUBE_LIB): is its K loop slower in cycles, or is the clock lower?  (GPU box)"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, ".")
import bench
from cubecl_amd import ElemType, Mi355Runtime, TensorHandle
from cubecl_amd import _native as N
client = Mi355Runtime.client(); lib, ctx = client.lib, client.ctx
ev = bench.Events(client)
clk = client.empty(2 * 8192)
p0, p1 = C.c_void_p(clk.device_ptr()), C.c_void_p(clk.device_ptr() + 8192)
out = []
for (m, n, k, reps) in ((2048, 2048, 2048, 400), (2048, 2048, 8192, 200)):
    a = TensorHandle.uniform(client, (m, k), ElemType.BF16, 1, 1, -1.0, 1.0); b = TensorHandle.uniform(client, (n, k), ElemType.BF16, 1, 2, -1.0, 1.0)
    c = client.empty(m * n * 2)
    d = bench.gemm_desc(N, m, n, k, N.DTYPE_BF16, N.DTYPE_BF16, trans_b=1, algo=3)
    call = lambda: lib.mi355_gemm(ctx, None, C.byref(d), a.device_ptr(), b.device_ptr(), c.device_ptr())
    for _ in range(100): call()
    lib.mi355_memset(ctx, None, C.c_void_p(clk.device_ptr()), 0, 2 * 8192)
    client.sync()
    lib.mi355_probe_clock(ctx, None, p0)
    ev.start()
    for _ in range(reps): call()
    ms = ev.stop_ms() / reps
    lib.mi355_probe_clock(ctx, None, p1)
    client.sync()
    t = np.frombuffer(client.read_one(clk), dtype=np.uint64).reshape(2, 512, 2).astype(np.float64)
    ok = (t[0, :, 1] > 0) & (t[1, :, 1] > t[0, :, 1]) & (t[1, :, 0] > t[0, :, 0])
    ghz = float(np.median((t[1, ok, 0] - t[0, ok, 0]) / (t[1, ok, 1] - t[0, ok, 1]) * 0.1))
    nk = k // 64
    out.append(f"{m}x{n}x{k}: {ms * 1e3:6.1f} us  clock {ghz:5.3f} GHz  -> {ms * 1e3 * ghz * 1e3 / nk:6.0f} cycles per K-tile (launch included)")
print(os.path.basename(os.environ.get("MI355CUBE_LIB", "product")), " | ".join(out), flush=True)
