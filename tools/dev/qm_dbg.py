"""Dev: where the persistent 16x16x32 kernel's row-major-rhs form differs from its [N][K] form (tile, row, column pattern of the mismatches)."""
import sys, numpy as np
sys.path.insert(0, '.')
from cubecl_amd import ElemType, Mi355Runtime, TensorHandle, ops
from cubecl_amd import _native as N
cl = Mi355Runtime.client()
for (m, n, batch, k) in [(1024, 512, 40, 1728), (1024, 512, 40, 960), (1024, 512, 40, 640)]:
    a = TensorHandle.uniform(cl, (batch, m, k), ElemType.BF16, 0x5EEDC0BE, 83, -1.0, 1.0)
    bh = np.random.default_rng(1).integers(0x3c00, 0x3f80, size=(k, n)).astype(np.uint16)
    b_kn = TensorHandle.from_numpy(cl, bh, ElemType.BF16)
    b_nk = TensorHandle.from_numpy(cl, np.ascontiguousarray(bh.T), ElemType.BF16)
    outs = []
    for handle, strides in ((b_kn, (0, n, 1)), (b_nk, (0, 1, k))):
        c = TensorHandle.new_contiguous((batch, m, n), cl.empty(batch * m * n * 2), ElemType.BF16)
        cl._s.check(cl.lib.mi355_memset(cl.ctx, None, c.device_ptr(), 0xEE, batch * m * n * 2))
        ops.matmul(cl, a, TensorHandle.new(handle.handle, (batch, k, n), strides, ElemType.BF16), c, algo=N.GEMM_ALGO_LP_256QM)
        outs.append(c.to_numpy(cl).copy())
    got, want = outs[0].view(np.uint16), outs[1].view(np.uint16)
    bad = np.argwhere(got != want)
    print((m, n, batch, k), 'mismatches', len(bad))
    if len(bad):
        bs, rs, cs = bad[:, 0], bad[:, 1], bad[:, 2]
        print('  batches', np.unique(bs)[:12], 'rows%256', np.unique(rs % 256)[:48], 'n', len(np.unique(rs % 256)))
        print('  cols%256', np.unique(cs % 256)[:64], 'n', len(np.unique(cs % 256)))
        print('  first', bad[0], 'got', hex(got[bs[0], rs[0], cs[0]]), 'want', hex(want[bs[0], rs[0], cs[0]]))
