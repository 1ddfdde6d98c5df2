import sys, numpy as np, ctypes as C
sys.path.insert(0, ".")
from cubecl_amd import ElemType, Mi355Runtime, TensorHandle, ops
from cubecl_amd import _native as N
client = Mi355Runtime.client(); lib, ctx = client.lib, client.ctx
for (m, n, k) in ((2176, 2176, 64), (2176, 2176, 256), (8192, 8192, 64), (2100, 2260, 128)):
    a = TensorHandle.uniform(client, (m, k), ElemType.BF16, 1, 1, -1.0, 1.0); b = TensorHandle.uniform(client, (n, k), ElemType.BF16, 1, 2, -1.0, 1.0)
    bt = TensorHandle.new(b.handle, (k, n), (1, k), ElemType.BF16)
    ref = TensorHandle.new_contiguous((m, n), client.empty(m * n * 2), ElemType.BF16)
    ops.matmul(client, a, bt, ref, algo=5)
    want = ref.to_numpy(client).copy()
    c = TensorHandle.new_contiguous((m, n), client.empty(m * n * 2), ElemType.BF16)
    fails = 0
    for it in range(60):
        lib.mi355_memset(ctx, None, C.c_void_p(c.device_ptr()), 0xEE, m * n * 2)
        ops.matmul(client, a, bt, c, algo=3)
        got = c.to_numpy(client)
        bad = np.argwhere(got.reshape(m, n) != want.reshape(m, n))
        if len(bad):
            fails += 1
            if fails <= 3:
                g = got.reshape(m, n)
                print(f"  it {it}: {len(bad)} bad; rows {sorted(set(bad[:,0].tolist()))[:8]} cols {sorted(set(bad[:,1].tolist()))[:12]} vals {[hex(int(g[i,j])) for i,j in bad[:4]]}")
    print(f"{m}x{n}x{k}: {fails} of 60 launches differ", flush=True)
