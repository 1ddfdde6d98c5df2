#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
cat > /tmp/c5ab.py <<'PY'
import ctypes as C, sys
sys.path.insert(0, ".")
import bench
from cubecl_amd import ElemType, Mi355Runtime, TensorHandle
from cubecl_amd import _native as N
client = Mi355Runtime.client(); lib, ctx = client.lib, client.ctx
ev = bench.Events(client)
B, M = 128, 2048
a = TensorHandle.uniform(client, (B, M, M), ElemType.BF16, 1, 1, -1.0, 1.0); b = TensorHandle.uniform(client, (B, M, M), ElemType.BF16, 1, 2, -1.0, 1.0)
c = client.empty(B * M * M * 2)
d = bench.gemm_desc(N, M, M, M, N.DTYPE_BF16, N.DTYPE_BF16, trans_b=1, batch=B)
call = lambda: client._s.check(lib.mi355_gemm(ctx, None, C.byref(d), a.device_ptr(), b.device_ptr(), c.device_ptr()))
bench.time_op(client, ev, call, 20)
ms = sorted(bench.time_op(client, ev, call, 20) for _ in range(3))[1]
print(f"C5 x{B}: {ms*1e3:.1f} us  {2.0*M**3*B/ms/1e9:.1f} TFLOP/s")
PY
MI355CUBE_LIB=$PWD/cubecl_amd/csrc/variants/libmi355cube_qs4.so timeout 600 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_full_size.py -m gpu -q --no-header -p no:cacheprovider -k "lp256q or c5 or dripped" 2>&1 | tail -3
for rep in 1 2 3; do for so in libmi355cube.so variants/libmi355cube_qs4.so; do
  echo -n "$so: "; MI355CUBE_LIB=$PWD/cubecl_amd/csrc/$so python /tmp/c5ab.py
done; done 2>&1 | tee gpurun_out/r03aq_c5_drip.txt
