"""dev: one line of TFLOP/s for the shapes the 256x256 kernel is tuned on (bf16 / fp8 / MXFP4)."""
import ctypes as C, json, sys
sys.path.insert(0, ".")
import bench
from cubecl_amd import ElemType, Mi355Runtime, TensorHandle
from cubecl_amd import _native as N
client = Mi355Runtime.client(); lib, ctx = client.lib, client.ctx
ev = bench.Events(client)
out = {}
def run(name, call, flop):
    bench.time_op(client, ev, call, 40)
    ms = bench.time_op(client, ev, call, 30)
    out[name] = round(flop / ms / 1e9)
for S in (8192, 4096):
    a = TensorHandle.uniform(client, (S, S), ElemType.BF16, 1, 1, -1.0, 1.0); b = TensorHandle.uniform(client, (S, S), ElemType.BF16, 1, 2, -1.0, 1.0)
    c = client.empty(S * S * 2)
    d = bench.gemm_desc(N, S, S, S, N.DTYPE_BF16, N.DTYPE_BF16)
    run(f"bf16_{S}", lambda: client._s.check(lib.mi355_gemm(ctx, None, C.byref(d), a.device_ptr(), b.device_ptr(), c.device_ptr())), 2.0 * S ** 3)
    a8 = TensorHandle.uniform(client, (S, S), ElemType.F8E4M3, 1, 3, -1.0, 1.0); b8 = TensorHandle.uniform(client, (S, S), ElemType.F8E4M3, 1, 4, -1.0, 1.0)
    d8 = bench.gemm_desc(N, S, S, S, N.DTYPE_F8E4M3, N.DTYPE_BF16)
    run(f"fp8_{S}", lambda: client._s.check(lib.mi355_gemm(ctx, None, C.byref(d8), a8.device_ptr(), b8.device_ptr(), c.device_ptr())), 2.0 * S ** 3)
    a4 = TensorHandle.uniform(client, (S * S // 2,), ElemType.F4E2M1X2, 1, 5, 0.0, 256.0); b4 = TensorHandle.uniform(client, (S * S // 2,), ElemType.F4E2M1X2, 1, 6, 0.0, 256.0)
    sa = TensorHandle.uniform(client, (S * S // 32,), ElemType.UE8M0, 1, 7, 124.0, 131.0); sb = TensorHandle.uniform(client, (S * S // 32,), ElemType.UE8M0, 1, 8, 124.0, 131.0)
    d4 = N.GemmScaledDesc(m=S, n=S, k=S, batch=1, lda=S, ldb=S, ldc=S, ld_sa=S // 32, ld_sb=S // 32, dtype_a=N.DTYPE_F4E2M1X2, dtype_b=N.DTYPE_F4E2M1X2, dtype_c=N.DTYPE_BF16, block=32)
    run(f"mxfp4_{S}", lambda: client._s.check(lib.mi355_gemm_scaled(ctx, None, C.byref(d4), a4.device_ptr(), sa.device_ptr(), b4.device_ptr(), sb.device_ptr(), c.device_ptr())), 2.0 * S ** 3)
M = 2048
ba = TensorHandle.uniform(client, (64, M, M), ElemType.BF16, 1, 9, -1.0, 1.0); bb = TensorHandle.uniform(client, (64, M, M), ElemType.BF16, 1, 10, -1.0, 1.0)
bc = client.empty(64 * M * M * 2)
db = bench.gemm_desc(N, M, M, M, N.DTYPE_BF16, N.DTYPE_BF16, batch=64)
run("bf16_2048x64", lambda: client._s.check(lib.mi355_gemm(ctx, None, C.byref(db), ba.device_ptr(), bb.device_ptr(), bc.device_ptr())), 2.0 * M ** 3 * 64)
M = 1024
ba = TensorHandle.uniform(client, (256, M, M), ElemType.BF16, 1, 9, -1.0, 1.0); bb = TensorHandle.uniform(client, (256, M, M), ElemType.BF16, 1, 10, -1.0, 1.0)
bc = client.empty(256 * M * M * 2)
dc = bench.gemm_desc(N, M, M, M, N.DTYPE_BF16, N.DTYPE_BF16, batch=256)
run("bf16_1024x256", lambda: client._s.check(lib.mi355_gemm(ctx, None, C.byref(dc), ba.device_ptr(), bb.device_ptr(), bc.device_ptr())), 2.0 * M ** 3 * 256)
S = 16384
a = TensorHandle.uniform(client, (S, S), ElemType.BF16, 1, 1, -1.0, 1.0); b = TensorHandle.uniform(client, (S, S), ElemType.BF16, 1, 2, -1.0, 1.0)
c = client.empty(S * S * 2)
d16 = bench.gemm_desc(N, S, S, S, N.DTYPE_BF16, N.DTYPE_BF16)
run("bf16_16384", lambda: client._s.check(lib.mi355_gemm(ctx, None, C.byref(d16), a.device_ptr(), b.device_ptr(), c.device_ptr())), 2.0 * S ** 3)
print(json.dumps(out))
