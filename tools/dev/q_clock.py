"""dev: sustained shader clock and TFLOP/s of lp256q / lp256p / lp256w4 on the C5 shard (is the chip at its power limit?)."""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, ".")
import bench
from cubecl_amd import ElemType, Mi355Runtime, TensorHandle
from cubecl_amd import _native as N
client = Mi355Runtime.client(); lib, ctx = client.lib, client.ctx
ev = bench.Events(client)
m = n = k = 2048; batch = 64
a = TensorHandle.uniform(client, (batch, m, k), ElemType.BF16, 1, 1, -1.0, 1.0); b = TensorHandle.uniform(client, (batch, n, k), ElemType.BF16, 1, 2, -1.0, 1.0)
c = client.empty(batch * m * n * 2)
clk = client.empty(2 * 8192)
p0, p1 = C.c_void_p(clk.device_ptr()), C.c_void_p(clk.device_ptr() + 8192)
def run(algo, reps=60):
    d = bench.gemm_desc(N, m, n, k, N.DTYPE_BF16, N.DTYPE_BF16, batch=batch, algo=algo)
    call = lambda: lib.mi355_gemm(ctx, None, C.byref(d), a.device_ptr(), b.device_ptr(), c.device_ptr())
    for _ in range(30): call()
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
    tf = 2.0 * m * n * k * batch / ms / 1e9
    return tf, ghz
for rep in range(3):
    for name, algo in (("q", N.GEMM_ALGO_LP_256Q), ("p", N.GEMM_ALGO_LP_256P), ("w4", N.GEMM_ALGO_LP_256W4)):
        tf, ghz = run(algo)
        print(f"{name:3s} {tf:7.0f} TF  clock {ghz:5.3f} GHz   TF per GHz {tf / ghz:6.0f}", flush=True)
