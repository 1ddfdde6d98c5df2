#!/usr/bin/env python3
"""Dev (GPU box): AUTO against every forced kernel on the GEMMs of two transformer layers (Llama-3-8B / -70B shaped: hidden 4096 / 8192,
FFN 14336 / 28672, fused QKV 6144 / 10240, vocabulary 128256) at 1 ... 8192 tokens, both rhs layouts ([N][K] = nn.Linear's weight as stored,
row-major [K][N] = what TensorHandle::new_contiguous gives a rhs), cold operands; prints every shape, flags AUTO more than 10 % and 2 us behind.
usage: tools/dev/llm_audit.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ab_algos, bench
from cubecl_amd import Mi355Runtime
cl = Mi355Runtime.client(); ev = bench.Events(cl)
TOKENS = [1, 4, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192]
LAYERS = {"8B": [(6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336)],          # qkv, o, gate+up, down: (out_features, in_features)
          "70B": [(10240, 8192), (8192, 8192), (57344, 8192), (8192, 28672)]}
shapes = []
for name, mats in LAYERS.items():
    for (n, k) in mats:
        for t in TOKENS:
            if 2.0 * t * n * k <= 4e12 and 2.0 * (t * k + n * k + t * n) <= 1.6e9:
                shapes.append((t, n, k))
shapes += [(t, 128256, 4096) for t in (1, 16, 64, 512)]                               # the LM head
shapes = sorted(set(shapes))
ALGOS = ["auto", "lp128", "lp256x128", "lp256w4", "lp256p", "lp256q", "stream64", "skinny", "lp256x192", "lp192x192", "lp256m16", "lp256qm"]
behind = 0
for nn in (False, True):
    algos = ALGOS + (["nnrows"] if nn else [])
    res = ab_algos.measure(cl, ev, shapes, algos, rounds=3, iters=10, nn=nn)
    print(f"== rhs {'row-major [K][N]' if nn else '[N][K]'}: {len(shapes)} shapes  (tokens x out_features x in_features)")
    for (m, n, k), r in res.items():
        us = {a: t for a, t in r["us"].items() if t == t}
        forced = [(a, t) for a, t in us.items() if a != "auto"]
        if not forced:
            print(f"{m:6d}x{n:6d}x{k:6d}: AUTO -> {r['auto']:9s} {us['auto']:8.1f} us   (no kernel takes it forced: re-laid out)", flush=True)
            continue
        best_a, best = min(forced, key=lambda x: x[1])
        ratio = us["auto"] / best
        flag = "  <-- BEHIND" if ratio > 1.10 and us["auto"] - best > 2.0 else ""
        behind += bool(flag)
        tf = 2.0 * m * n * k / us["auto"] / 1e6
        print(f"{m:6d}x{n:6d}x{k:6d}: AUTO -> {r['auto']:9s} {us['auto']:8.1f} us {tf:7.0f} TFLOP/s   best {best_a:9s} {best:8.1f} us   x{ratio:.3f}{flag}", flush=True)
print(f"{behind} of {2 * len(shapes)} more than 10 % + 2 us behind")
