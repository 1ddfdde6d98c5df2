"""dev: AUTO kernel choice and TFLOP/s over mid-size bf16 shapes."""
import ctypes as C, json, sys
sys.path.insert(0, ".")
import bench
from cubecl_amd import ElemType, Mi355Runtime, TensorHandle
from cubecl_amd import _native as N
client = Mi355Runtime.client(); lib, ctx = client.lib, client.ctx
ev = bench.Events(client)
SHAPES = [tuple(map(int, x.split("x"))) for x in sys.argv[1:] if x[0].isdigit()] if (len(sys.argv) > 1 and sys.argv[1] not in ("f32", "fp8")) else [(2048, 2048, 2048)]
names = {1: "generic", 2: "f32", 3: "lp128", 4: "lp256", 5: "w4", 6: "p"}
for (m, n, k) in SHAPES:
    if len(sys.argv) > 1 and sys.argv[1] in ("f32", "fp8"):
        break
    a = TensorHandle.uniform(client, (m, k), ElemType.BF16, 1, 1, -1.0, 1.0); b = TensorHandle.uniform(client, (n, k), ElemType.BF16, 1, 2, -1.0, 1.0)
    c = client.empty(m * n * 2)
    res = {}
    for algo in (0, N.GEMM_ALGO_LP_128, N.GEMM_ALGO_LP_256W4):
        d = N.GemmDesc(m=m, n=n, k=k, batch=1, lda=k, ldb=k, ldc=n, dtype_ab=N.DTYPE_BF16, dtype_c=N.DTYPE_BF16, trans_b=1, algo=algo)
        call = lambda: client._s.check(lib.mi355_gemm(ctx, None, C.byref(d), a.device_ptr(), b.device_ptr(), c.device_ptr()))
        try:
            bench.time_op(client, ev, call, 30)
            ms = bench.time_op(client, ev, call, 30)
            sel = C.c_int32(algo)
            if algo == 0:
                lib.mi355_gemm_select(ctx, C.byref(d), C.byref(sel))
            res[("auto=" + names.get(sel.value, "?")) if algo == 0 else names[algo]] = round(2.0 * m * n * k / ms / 1e9)
        except Exception as e:
            res[names[algo]] = "n/a"
    print(f"{m}x{n}x{k}", res)

if len(sys.argv) > 1 and sys.argv[1] == "fp8":
    for (m, n, k) in [tuple(map(int, x.split("x"))) for x in sys.argv[2:]]:
        a = TensorHandle.uniform(client, (m, k), ElemType.F8E4M3, 1, 1, -1.0, 1.0); b = TensorHandle.uniform(client, (n, k), ElemType.F8E4M3, 1, 2, -1.0, 1.0)
        c = client.empty(m * n * 2)
        res = {}
        for algo in (0, N.GEMM_ALGO_LP_128, N.GEMM_ALGO_LP_256W4):
            d = N.GemmDesc(m=m, n=n, k=k, batch=1, lda=k, ldb=k, ldc=n, dtype_ab=N.DTYPE_F8E4M3, dtype_c=N.DTYPE_BF16, trans_b=1, algo=algo)
            call = lambda: client._s.check(lib.mi355_gemm(ctx, None, C.byref(d), a.device_ptr(), b.device_ptr(), c.device_ptr()))
            bench.time_op(client, ev, call, 30)
            ms = bench.time_op(client, ev, call, 30)
            sel = C.c_int32(algo)
            if algo == 0:
                lib.mi355_gemm_select(ctx, C.byref(d), C.byref(sel))
            res[("auto=" + names.get(sel.value, "?")) if algo == 0 else names[algo]] = round(2.0 * m * n * k / ms / 1e9)
        print("fp8", f"{m}x{n}x{k}", res)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "f32":
    for (m, n, k) in [tuple(map(int, x.split("x"))) for x in sys.argv[2:]]:
        a = TensorHandle.uniform(client, (m, k), ElemType.F32, 1, 1, -1.0, 1.0); b = TensorHandle.uniform(client, (n, k), ElemType.F32, 1, 2, -1.0, 1.0)
        c = client.empty(m * n * 4)
        res = {}
        for algo in (0, N.GEMM_ALGO_F32_MFMA, N.GEMM_ALGO_LP_256W4):
            d = N.GemmDesc(m=m, n=n, k=k, batch=1, lda=k, ldb=k, ldc=n, dtype_ab=N.DTYPE_F32, dtype_c=N.DTYPE_F32, trans_b=1, algo=algo)
            call = lambda: client._s.check(lib.mi355_gemm(ctx, None, C.byref(d), a.device_ptr(), b.device_ptr(), c.device_ptr()))
            bench.time_op(client, ev, call, 10)
            ms = bench.time_op(client, ev, call, 10)
            sel = C.c_int32(algo)
            if algo == 0:
                lib.mi355_gemm_select(ctx, C.byref(d), C.byref(sel))
            res[("auto=" + names.get(sel.value, "?")) if algo == 0 else names[algo]] = round(2.0 * m * n * k / ms / 1e9, 1)
        print("f32", f"{m}x{n}x{k}", res)
