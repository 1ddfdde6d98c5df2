"""dev: which 16-byte chunk of a K-tile row does the MX kernel scale with which scale byte?"""
import sys
import numpy as np
sys.path.insert(0, ".")
from cubecl_amd import ElemType, Mi355Runtime, TensorHandle, ops
from cubecl_amd import _native as N

client = Mi355Runtime.client()
U = ElemType.UE8M0
for da, one, two, kb in ((ElemType.F8E4M3, 0x38, 0x40, 1), (ElemType.F4E2M1X2, 0x22, 0x44, 2)):
    m = n = 256
    k = 512 * kb
    nb = k // 32
    nblk_tile = 4 * kb
    ones = np.full((m, nb), 127, dtype=np.uint8)
    b = np.full((n, k // kb), one, dtype=np.uint8)
    tb = TensorHandle.from_numpy(client, b, da)
    c = TensorHandle.new_contiguous((m, n), client.empty(m * n * 4), ElemType.F32)
    print(da.name, "rows: 16-byte chunk of K-tile 0 holding 2.0; cols: block whose A scale is 2^10; entry 1 = that chunk is scaled by that block")
    for ch in range(8):
        a = np.full((m, k // kb), one, dtype=np.uint8)
        a[:, ch * 16:(ch + 1) * 16] = two
        ta = TensorHandle.from_numpy(client, a, da)
        row = []
        for blk in range(nblk_tile):
            sa = ones.copy()
            sa[:, blk] = 137
            ops.matmul_scaled(client, ta, TensorHandle.from_numpy(client, sa, U), ta if False else tb, TensorHandle.from_numpy(client, ones, U), c,
                              algo=N.GEMM_ALGO_LP_256W4)
            g = c.to_numpy(client)
            assert np.all(g == g[0, 0])
            epc = 16 * kb                      # elements per 16-byte chunk
            per_blk_chunks = 32 // epc         # chunks per MX block: 2 (fp8) / 1 (fp4)
            total_chunks = k // epc
            hit = epc * (2 * 1024 + (per_blk_chunks - 1) * 1024 + (total_chunks - per_blk_chunks))
            miss = epc * (2 + per_blk_chunks * 1024 + (total_chunks - 1 - per_blk_chunks))
            row.append(1 if g[0, 0] == hit else 0 if g[0, 0] == miss else -1)
        print("  chunk", ch, row)
