"""dev: cost of D = A*B + C over plain C = A*B."""
import ctypes as C, sys
sys.path.insert(0, ".")
import bench
from cubecl_amd import ElemType, Mi355Runtime, TensorHandle
from cubecl_amd import _native as N
client = Mi355Runtime.client(); lib, ctx = client.lib, client.ctx
ev = bench.Events(client)
for (m, n, k, dt, odt) in ((8192, 8192, 8192, ElemType.BF16, ElemType.BF16), (8192, 8192, 8192, ElemType.BF16, ElemType.F32),
                           (4096, 4096, 4096, ElemType.F32, ElemType.F32), (4096, 4096, 4096, ElemType.BF16, ElemType.BF16),
                           (8192, 8192, 1024, ElemType.BF16, ElemType.BF16)):
    a = TensorHandle.uniform(client, (m, k), dt, 1, 1, -1.0, 1.0)
    b = TensorHandle.uniform(client, (n, k), dt, 1, 2, -1.0, 1.0)
    c = TensorHandle.uniform(client, (m, n), odt, 1, 3, -1.0, 1.0)
    d = client.empty(m * n * odt.size())
    desc = bench.gemm_desc(N, m, n, k, int(dt), int(odt), trans_b=1)
    for _ in range(40):
        lib.mi355_gemm(ctx, None, C.byref(desc), a.device_ptr(), b.device_ptr(), d.device_ptr())
    t0, _ = bench.samples_op(client, ev, lambda: client._s.check(lib.mi355_gemm(ctx, None, C.byref(desc), a.device_ptr(), b.device_ptr(), d.device_ptr())))
    t1, _ = bench.samples_op(client, ev, lambda: client._s.check(lib.mi355_gemm_add(ctx, None, C.byref(desc), a.device_ptr(), b.device_ptr(), c.device_ptr(), d.device_ptr())))
    print(f"{m}x{n}x{k} {dt.name}->{odt.name}: gemm {t0*1e3:8.1f} us  gemm_add {t1*1e3:8.1f} us  (+{(t1/t0-1)*100:.1f} %)  {2.0*m*n*k/t1/1e9:7.1f} TF", flush=True)
