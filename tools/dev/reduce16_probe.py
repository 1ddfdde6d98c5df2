"""dev: array-wide reductions, f32 vs bf16 input, GB/s and elements/s."""
import ctypes as C, sys
sys.path.insert(0, ".")
import bench
from cubecl_amd import ElemType, Mi355Runtime, TensorHandle
from cubecl_amd import _native as N
client = Mi355Runtime.client(); lib, ctx = client.lib, client.ctx
ev = bench.Events(client)
ws = client.empty(1 << 17); o = client.empty(64)
for dt, esz in ((ElemType.F32, 4), (ElemType.BF16, 2), (ElemType.F16, 2)):
    n = (1 << 30) // esz
    x = TensorHandle.uniform(client, (n,), dt, 1, 300, 0.0, 1.0)
    for name, fn in (("sum", lambda: lib.mi355_reduce_sum(ctx, None, x.device_ptr(), int(dt), n, o.device_ptr(), ws.device_ptr(), ws.size)),
                     ("argmax", lambda: lib.mi355_argmax(ctx, None, x.device_ptr(), int(dt), n, o.device_ptr(), o.device_ptr() + 8, ws.device_ptr(), ws.size)),
                     ("fused", lambda: lib.mi355_sum_argmax(ctx, None, x.device_ptr(), int(dt), n, o.device_ptr(), o.device_ptr() + 16, o.device_ptr() + 8, ws.device_ptr(), ws.size))):
        med, best = bench.samples_op(client, ev, lambda: client._s.check(fn()))
        print(f"{dt.name:5s} {name:7s} 1 GiB: median {med * 1e3:7.1f} us  {(1 << 30) / med / 1e6:7.0f} GB/s  min {(1 << 30) / best / 1e6:7.0f} GB/s  {n / med / 1e6:8.0f} Gelem/s")
