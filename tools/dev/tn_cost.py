"""What does a transposed lhs cost today?  C[M][N] = A^T B with A stored [K][M] (trans_a) and B row-major [K][N] -- the
weight-gradient product of a training step (lhs^T . grad_out) -- against the same shape with both operands K-contiguous."""
import ctypes as C
import statistics
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import bench
from cubecl_amd import ElemType, Mi355Runtime, TensorHandle
from cubecl_amd import _native as N

client = Mi355Runtime.client()
lib, ctx = client.lib, client.ctx
ev = bench.Events(client)
for (m, n, k) in [(4096, 4096, 8192), (8192, 8192, 8192), (2048, 2048, 8192), (1024, 4096, 16384), (4096, 1024, 4096), (512, 512, 8192)]:
    fp = 2 * (m * k + n * k + m * n)
    nsets = max(1, min(8, -(-(768 << 20) // fp)))
    sets = [(TensorHandle.uniform(client, (m, k), ElemType.BF16, 1, 2 * i + 1, -1.0, 1.0), TensorHandle.uniform(client, (n, k), ElemType.BF16, 1, 2 * i + 2, -1.0, 1.0),
             client.empty(m * n * 2)) for i in range(nsets)]
    row = []
    for (ta, tb) in ((0, 1), (0, 0), (1, 0), (1, 1)):
        d = N.GemmDesc(m=m, n=n, k=k, batch=1, lda=(m if ta else k), ldb=(k if tb else n), ldc=n, dtype_ab=N.DTYPE_BF16, dtype_c=N.DTYPE_BF16, trans_a=ta, trans_b=tb, algo=0)
        turn = [0]

        def call():
            sa, sb, sc = sets[turn[0] % nsets]
            turn[0] += 1
            rc = lib.mi355_gemm(ctx, None, C.byref(d), sa.device_ptr(), sb.device_ptr(), sc.device_ptr())
            assert rc == N.OK, rc
        t = statistics.median(bench.time_op(client, ev, call, 20, warmup=3) * 1e3 for _ in range(3))
        row.append(f"trans_a={ta} trans_b={tb}: {t:8.1f} us {2.0 * m * n * k / t / 1e6:7.0f} TF")
    print(f"{m}x{n}x{k}  " + " | ".join(row), flush=True)
