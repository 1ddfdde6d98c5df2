"""dev: AUTO against every forced kernel over a grid of bf16 shapes; prints the shapes where AUTO is more than 8 % behind the
best forced choice (GPU box).  usage: python tools/dev/select_audit.py [seed]"""
import ctypes as C, itertools, random, sys
sys.path.insert(0, ".")
import bench
from cubecl_amd import ElemType, Mi355Runtime, TensorHandle
from cubecl_amd import _native as N
client = Mi355Runtime.client(); lib, ctx = client.lib, client.ctx
ev = bench.Events(client)
NAMES = {0: "auto", 3: "lp128", 5: "w4", 6: "p", 7: "q", 8: "skinny", 9: "stream64"}
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
MS = [1, 2, 4, 16, 32, 48, 64, 96, 128, 192, 256, 384, 512, 1024, 2048, 3072, 4096, 8192]
KS = [64, 128, 256, 512, 1024, 2048, 4096, 8192, 14336]
shapes = set()
while len(shapes) < 70:
    m, n, k = rng.choice(MS), rng.choice(MS + [6144, 14336, 28672]), rng.choice(KS)
    if rng.random() < 0.5:
        m, n = n, m
    if m * n * 2 > (1 << 29) or (m * k + n * k) * 2 > (1 << 30) or m * n * k < (1 << 22):
        continue
    shapes.add((m, n, k))
bad = 0
for (m, n, k) in sorted(shapes):
    a = TensorHandle.uniform(client, (m, k), ElemType.BF16, 1, 1, -1.0, 1.0); b = TensorHandle.uniform(client, (n, k), ElemType.BF16, 1, 2, -1.0, 1.0)
    c = client.empty(m * n * 2)
    res = {}
    for algo in (0, 3, 5, 6, 7, 8, 9):
        d = bench.gemm_desc(N, m, n, k, N.DTYPE_BF16, N.DTYPE_BF16, trans_b=1, algo=algo)
        if lib.mi355_gemm(ctx, None, C.byref(d), a.device_ptr(), b.device_ptr(), c.device_ptr()) != 0:
            continue
        res[algo] = min(bench.time_op(client, ev, lambda: lib.mi355_gemm(ctx, None, C.byref(d), a.device_ptr(), b.device_ptr(), c.device_ptr()), 10, warmup=2) for _ in range(3)) * 1e3
    client.sync()
    d = bench.gemm_desc(N, m, n, k, N.DTYPE_BF16, N.DTYPE_BF16, trans_b=1, algo=0)
    sel = C.c_int32(); lib.mi355_gemm_select(ctx, C.byref(d), C.byref(sel))
    best = min((v, a_) for a_, v in res.items() if a_ != 0)
    flag = res[0] > 1.08 * best[0] and res[0] - best[0] > 1.0
    bad += flag
    if flag or "-v" in sys.argv:
        print(f"{m}x{n}x{k}: auto->{NAMES.get(sel.value, sel.value)} {res[0]:7.1f} us   best {NAMES[best[1]]} {best[0]:7.1f} us   " +
              "  ".join(f"{NAMES[a_]} {v:.1f}" for a_, v in sorted(res.items()) if a_ != 0), flush=True)
print(f"{len(shapes)} shapes, {bad} where AUTO is more than 8 % (and 1 us) behind the best forced kernel")
