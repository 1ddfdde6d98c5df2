"""dev: fp8 GEMM + matrix-pipe ceilings (not part of the product or the tests)."""
import ctypes as C
import json
import sys

sys.path.insert(0, ".")
import bench
from cubecl_amd import ElemType, Mi355Runtime, TensorHandle
from cubecl_amd import _native as N

client = Mi355Runtime.client()
lib, ctx = client.lib, client.ctx
ev = bench.Events(client)
sink = client.empty(256)
out = {}
n_ops = C.c_uint64()
for name, fn in (("mfma_fp8_ones", lambda: lib.mi355_probe_mfma(ctx, None, N.DTYPE_F8E4M3, 10000, sink.device_ptr(), C.byref(n_ops))),
                 ("mfma_fp8_uniform", lambda: lib.mi355_probe_mfma_data(ctx, None, 2, 10000, sink.device_ptr(), C.byref(n_ops))),
                 ("mfma_bf16_uniform", lambda: lib.mi355_probe_mfma_data(ctx, None, 1, 20000, sink.device_ptr(), C.byref(n_ops)))):
    ms = bench.time_op(client, ev, lambda: client._s.check(fn()), 5)
    out[name] = round(n_ops.value / ms / 1e9, 1)
for dt, dn in ((ElemType.F8E4M3, "e4m3"), (ElemType.F8E5M2, "e5m2")):
    for S in (8192, 4096, 2048):
        a = TensorHandle.uniform(client, (S, S), dt, 1, 900, -1.0, 1.0)
        b = TensorHandle.uniform(client, (S, S), dt, 1, 901, -1.0, 1.0)
        c = client.empty(S * S * 2)
        d = bench.gemm_desc(N, S, S, S, int(dt), N.DTYPE_BF16, trans_b=1)
        call = lambda: client._s.check(lib.mi355_gemm(ctx, None, C.byref(d), a.device_ptr(), b.device_ptr(), c.device_ptr()))
        bench.time_op(client, ev, call, 40)
        ms = bench.time_op(client, ev, call, 30)
        out[f"gemm_{dn}_{S}"] = {"ms": round(ms, 4), "TFLOPs": round(2.0 * S ** 3 / ms / 1e9, 1)}
# bf16 for comparison on the same box
S = 8192
a = TensorHandle.uniform(client, (S, S), ElemType.BF16, 1, 900, -1.0, 1.0)
b = TensorHandle.uniform(client, (S, S), ElemType.BF16, 1, 901, -1.0, 1.0)
c = client.empty(S * S * 2)
d = bench.gemm_desc(N, S, S, S, N.DTYPE_BF16, N.DTYPE_BF16, trans_b=1)
call = lambda: client._s.check(lib.mi355_gemm(ctx, None, C.byref(d), a.device_ptr(), b.device_ptr(), c.device_ptr()))
bench.time_op(client, ev, call, 40)
ms = bench.time_op(client, ev, call, 30)
out["gemm_bf16_8192"] = {"ms": round(ms, 4), "TFLOPs": round(2.0 * S ** 3 / ms / 1e9, 1)}
print(json.dumps(out, indent=1))
