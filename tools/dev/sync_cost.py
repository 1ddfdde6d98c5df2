"""dev: where the fixed ~0.5 ms of bench.py's timed region goes (GPU box)."""
import ctypes as C, sys, time
sys.path.insert(0, ".")
import torch
import bench
from cubecl_amd import ElemType, Mi355Runtime, TensorHandle
from cubecl_amd import _native as N
client = Mi355Runtime.client(); lib, ctx = client.lib, client.ctx
ev = bench.Events(client)
S = 8192
a = TensorHandle.uniform(client, (S, S), ElemType.BF16, 1, 1, -1.0, 1.0); b = TensorHandle.uniform(client, (S, S), ElemType.BF16, 1, 2, -1.0, 1.0)
c = client.empty(S * S * 2)
d = bench.gemm_desc(N, S, S, S, N.DTYPE_BF16, N.DTYPE_BF16, trans_b=1)
step = lambda: lib.mi355_gemm(ctx, None, C.byref(d), a.device_ptr(), b.device_ptr(), c.device_ptr())
clk = client.empty(16384)
for _ in range(60): step()
client.sync()
pc = time.perf_counter
for trial in range(3):
    torch.cuda.synchronize(); client.sync()
    t0 = pc(); lib.mi355_probe_clock(ctx, None, C.c_void_p(clk.device_ptr())); t1 = pc()
    ev.start(); t2 = pc()
    for _ in range(20): step()
    t3 = pc()
    ms = ev.stop_ms(); t4 = pc()
    lib.mi355_probe_clock(ctx, None, C.c_void_p(clk.device_ptr() + 8192)); t5 = pc()
    torch.cuda.synchronize(); t6 = pc()
    client.sync(); t7 = pc()
    print(f"probe0 {1e3*(t1-t0):.3f}  ev.start {1e3*(t2-t1):.3f}  20 launches {1e3*(t3-t2):.3f}  ev.stop(sync) {1e3*(t4-t3):.3f}  probe1 {1e3*(t5-t4):.3f}  torch.sync {1e3*(t6-t5):.3f}  client.sync {1e3*(t7-t6):.3f}  | total {1e3*(t7-t0):.3f} ms, events {ms:.3f} ms, overhead {1e3*(t7-t0)-ms:.3f}")
