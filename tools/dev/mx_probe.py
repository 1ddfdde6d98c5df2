"""dev: block-scaled GEMM throughput (not part of the product or the tests)."""
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
out = {}
sink = client.empty(256)
n_ops = C.c_uint64()
for name, mode in (("pipe_mxfp4_ones", 3), ("pipe_mxfp4_random", 4), ("pipe_fp8_uniform", 2)):
    ms = bench.time_op(client, ev, lambda: client._s.check(lib.mi355_probe_mfma_data(ctx, None, mode, 20000, sink.device_ptr(), C.byref(n_ops))), 5)
    out[name] = round(n_ops.value / ms / 1e9, 1)
for dt, dn, epb in ((ElemType.F8E4M3, "mxfp8_e4m3", 1), (ElemType.F4E2M1X2, "mxfp4", 2)):
    for S in (8192, 16384, 4096):
        if S == 16384 and epb == 1:
            continue
        nbytes = S * S // epb
        if epb == 1:
            a = TensorHandle.uniform(client, (S, S), dt, 1, 900, -1.0, 1.0)
            b = TensorHandle.uniform(client, (S, S), dt, 1, 901, -1.0, 1.0)
        else:      # random nibble pairs
            a = TensorHandle.uniform(client, (nbytes,), dt, 1, 900, 0.0, 256.0)
            b = TensorHandle.uniform(client, (nbytes,), dt, 1, 901, 0.0, 256.0)
        sa = TensorHandle.uniform(client, (S * S // 32,), ElemType.UE8M0, 1, 902, 124.0, 131.0)
        sb = TensorHandle.uniform(client, (S * S // 32,), ElemType.UE8M0, 1, 903, 124.0, 131.0)
        c = client.empty(S * S * 2)
        d = N.GemmScaledDesc(m=S, n=S, k=S, batch=1, lda=S, ldb=S, ldc=S, ld_sa=S // 32, ld_sb=S // 32, dtype_a=int(dt), dtype_b=int(dt),
                             dtype_c=N.DTYPE_BF16, block=32)
        call = lambda: client._s.check(lib.mi355_gemm_scaled(ctx, None, C.byref(d), a.device_ptr(), sa.device_ptr(), b.device_ptr(),
                                                             sb.device_ptr(), c.device_ptr()))
        bench.time_op(client, ev, call, 40)
        ms = bench.time_op(client, ev, call, 30)
        out[f"{dn}_{S}"] = {"ms": round(ms, 4), "TFLOPs": round(2.0 * S ** 3 / ms / 1e9, 1)}
print(json.dumps(out, indent=1))
