"""dev: the 128x128 kernel (algo 3) on mid-size shapes for the library named by MI355CUBE_LIB: time per launch and,
for builds that are meant to be correct (CHECK=1), the largest difference from the 256x256 kernel's result (GPU box)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, ".")
import bench
from cubecl_amd import ElemType, Mi355Runtime, TensorHandle
from cubecl_amd import _native as N
client = Mi355Runtime.client(); lib, ctx = client.lib, client.ctx
ev = bench.Events(client)
check = os.environ.get("CHECK", "0") == "1"
out = []
for (m, n, k) in ((2048, 2048, 2048), (2048, 2048, 8192), (1024, 4096, 4096), (4096, 2048, 2048), (4096, 4096, 1024), (2560, 2560, 2560), (8192, 8192, 64), (8192, 8192, 256)):
    a = TensorHandle.uniform(client, (m, k), ElemType.BF16, 1, 1, -1.0, 1.0); b = TensorHandle.uniform(client, (n, k), ElemType.BF16, 1, 2, -1.0, 1.0)
    c = client.empty(m * n * 2)
    d = bench.gemm_desc(N, m, n, k, N.DTYPE_BF16, N.DTYPE_BF16, trans_b=1, algo=3)
    run = lambda: lib.mi355_gemm(ctx, None, C.byref(d), a.device_ptr(), b.device_ptr(), c.device_ptr())
    assert run() == 0
    best = min(bench.time_op(client, ev, run, 20, warmup=3) for _ in range(5))
    txt = f"{m}x{n}x{k} {best * 1e3:6.1f}us"
    if check:
        got = np.frombuffer(client.read_one(c), dtype=np.uint16).astype(np.uint32) << 16
        d5 = bench.gemm_desc(N, m, n, k, N.DTYPE_BF16, N.DTYPE_BF16, trans_b=1, algo=5)
        assert lib.mi355_gemm(ctx, None, C.byref(d5), a.device_ptr(), b.device_ptr(), c.device_ptr()) == 0
        ref = np.frombuffer(client.read_one(c), dtype=np.uint16).astype(np.uint32) << 16
        diff = np.abs(got.view(np.float32) - ref.view(np.float32)).max()
        txt += f" maxdiff {diff:.3g} same_bits {bool((got == ref).all())}"
    out.append(txt)
print(os.path.basename(os.environ.get("MI355CUBE_LIB", "product")), " | ".join(out), flush=True)
