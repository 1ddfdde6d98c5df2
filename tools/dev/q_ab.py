"""dev: the dripped-epilogue persistent kernel (lp256q) against lp256p and lp256w4 -- bit-for-bit first, then interleaved timing.
usage: python tools/dev/q_ab.py [MxNxK[xBATCH] ...]"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, ".")
import bench
from cubecl_amd import ElemType, Mi355Runtime, TensorHandle
from cubecl_amd import _native as N
client = Mi355Runtime.client(); lib, ctx = client.lib, client.ctx
ev = bench.Events(client)
specs = sys.argv[1:] or ["2048x2048x2048x64", "2048x2048x1024x64", "2048x2048x640x64", "2048x2048x448x64", "2048x2048x384x64",
                         "2048x2048x2048x5", "2048x2048x2048x3", "8192x8192x8192", "8192x8192x4096", "4096x4096x4096x4", "2048x2048x512x64"]
ALGOS = (("q", N.GEMM_ALGO_LP_256Q), ("p", N.GEMM_ALGO_LP_256P), ("w4", N.GEMM_ALGO_LP_256W4))
for spec in specs:
    parts = list(map(int, spec.split("x")))
    m, n, k = parts[:3]; batch = parts[3] if len(parts) > 3 else 1
    a = TensorHandle.uniform(client, (batch, m, k), ElemType.BF16, 1, 1, -1.0, 1.0); b = TensorHandle.uniform(client, (batch, n, k), ElemType.BF16, 1, 2, -1.0, 1.0)
    outs = {name: client.empty(batch * m * n * 2) for name, _ in ALGOS}
    calls, ok = {}, {}
    for name, algo in ALGOS:
        d = bench.gemm_desc(N, m, n, k, N.DTYPE_BF16, N.DTYPE_BF16, batch=batch, algo=algo)
        calls[name] = (lambda d=d, o=outs[name]: lib.mi355_gemm(ctx, None, C.byref(d), a.device_ptr(), b.device_ptr(), o.device_ptr()))
        lib.mi355_memset(ctx, None, outs[name].device_ptr(), 0xEE, outs[name].size)
        ok[name] = calls[name]() == 0
    client.sync()
    ref = np.frombuffer(client.read_one(outs["w4"]), dtype=np.uint16)
    verdict = {}
    for name in ("q", "p"):
        if not ok[name]:
            verdict[name] = "unsupported"; continue
        got = np.frombuffer(client.read_one(outs[name]), dtype=np.uint16)
        bad = int(np.count_nonzero(got != ref))
        verdict[name] = "bit-exact" if bad == 0 else f"MISMATCH {bad} of {got.size} (first at {int(np.flatnonzero(got != ref)[0])})"
    # race screen: 5 more launches of q must reproduce the bits
    if ok["q"]:
        for _ in range(5):
            calls["q"]()
        client.sync()
        again = np.frombuffer(client.read_one(outs["q"]), dtype=np.uint16)
        if np.count_nonzero(again != ref):
            verdict["q"] += " / UNSTABLE under repetition"
    bench.time_op(client, ev, calls["w4"], 40)
    res = {name: [] for name, _ in ALGOS if ok[name]}
    for rep in range(4):
        for name in res:
            ms = bench.time_op(client, ev, calls[name], 20, warmup=1)
            res[name].append(2.0 * m * n * k * batch / ms / 1e9)
    med = {name: sorted(v)[len(v) // 2] for name, v in res.items()}
    line = "  ".join(f"{name} {med[name]:6.0f}" for name in med)
    gain = f"  q/p {100 * (med['q'] / med['p'] - 1):+5.1f} %  q/w4 {100 * (med['q'] / med['w4'] - 1):+5.1f} %" if "q" in med and "p" in med else ""
    print(f"{spec:>20}  {line}{gain}   q: {verdict.get('q')}  p: {verdict.get('p')}", flush=True)
