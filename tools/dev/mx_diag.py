"""dev: which scale byte does each lane of the MX kernel apply?  (all-ones operands, structured scales)"""
import sys
import numpy as np
sys.path.insert(0, ".")
from cubecl_amd import ElemType, Mi355Runtime, TensorHandle, ops
from cubecl_amd import _native as N

client = Mi355Runtime.client()
U = ElemType.UE8M0
for da, one, kb in ((ElemType.F8E4M3, 0x38, 1), (ElemType.F4E2M1X2, 0x22, 2)):
    m = n = 256
    k = 512 * kb
    nb = k // 32
    a = np.full((m, k // kb), one, dtype=np.uint8)
    ta = TensorHandle.from_numpy(client, a, da)
    c = TensorHandle.new_contiguous((m, n), client.empty(m * n * 4), ElemType.F32)

    def run(sa, sb):
        ops.matmul_scaled(client, ta, TensorHandle.from_numpy(client, sa, U), ta, TensorHandle.from_numpy(client, sb, U), c,
                          algo=N.GEMM_ALGO_LP_256W4)
        return c.to_numpy(client).copy()
    ones = np.full((m, nb), 127, dtype=np.uint8)
    g = run(ones, ones)
    print(da.name, "unit scales: all ==k?", bool(np.all(g == k)), g[0, :4], g[100, 200])
    # A scale depends on the row only
    sa = (127 + (np.arange(m)[:, None] % 5) + 0 * np.arange(nb)[None, :]).astype(np.uint8)
    g = run(sa, ones)
    want = k * 2.0 ** (sa[:, :1].astype(np.float64) - 127) * np.ones((1, n))
    bad = np.argwhere(g != want)
    print(" A by row: mismatches", len(bad), "first", bad[:5].tolist(), [np.log2(g[i, j] / k) for i, j in bad[:5]])
    # A scale depends on the block only: block b has scale 2^(b) for b < nb
    sa = (127 + np.arange(nb)[None, :] % 8 + 0 * np.arange(m)[:, None]).astype(np.uint8)
    g = run(sa, ones)
    want = 32.0 * np.sum(2.0 ** (sa[0].astype(np.float64) - 127))
    print(" A by block: uniform?", bool(np.all(g == g[0, 0])), g[0, 0], "want", want)
    # a single block carries 2^20: which outputs see it, and how often
    for blk in (0, 1, 2, 3, 5, nb - 1):
        sa = np.full((m, nb), 127, dtype=np.uint8)
        sa[:, blk] = 147
        g = run(sa, ones)
        cnt = (g - (k - 32)) / 2.0 ** 20 / 32
        print("  block", blk, "-> contribution count per output (want 1.0):", np.unique(np.round(cnt[:, 0], 3))[:6])
    # same for B
    sb = (127 + (np.arange(n)[:, None] % 5) + 0 * np.arange(nb)[None, :]).astype(np.uint8)
    g = run(ones, sb)
    want = k * (2.0 ** (sb[:, :1].astype(np.float64) - 127)).T * np.ones((m, 1))
    print(" B by row: mismatches", int(np.sum(g != want)))
