import ctypes as C, os, sys
sys.path.insert(0, ".")
import bench
from cubecl_amd import ElemType, Mi355Runtime, TensorHandle, ops
cl = Mi355Runtime.client(); ev = bench.Events(cl)
CASES = [((8192, 8192), 0), ((16384, 16384), 0), ((4, 65536, 1024), 1), ((64, 256, 1024), 0), ((64, 256, 1024), 1), ((64, 64, 4096), 1), ((512, 8192), 0), ((2048, 2048), 0), ((16, 4096, 4096), 1)]
line = f"wg/cu {os.environ.get('MI355_AXIS_WG_PER_CU', '4'):>2s}: "
for shape, axis in CASES:
    x = TensorHandle.uniform(cl, shape, ElemType.F32, 1, 900, -1.0, 1.0)
    n = 1
    for d in shape: n *= d
    m = n // shape[axis]
    o = TensorHandle.new_contiguous((m,), cl.empty(m * 4), ElemType.F32)
    med, best = bench.samples_op(cl, ev, lambda: ops.reduce_axis(cl, x, o, axis, "sum"))
    line += f"{'x'.join(map(str, shape))}/{axis} {med*1e3:6.1f}  "
    del x, o
print(line, flush=True)
