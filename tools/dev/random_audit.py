#!/usr/bin/env python3
"""Dev (GPU box): AUTO against every forced kernel on seeded random bf16 shapes (both rhs layouts), cold operands; prints the
shapes where AUTO is more than 10 % and 2 us behind.  usage: tools/dev/random_audit.py [seed] [count]"""
import math, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ab_algos, bench
from cubecl_amd import Mi355Runtime
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = random.Random(seed)
cl = Mi355Runtime.client(); ev = bench.Events(cl)
def dim(lo, hi, mult):
    v = int(round(math.exp(rng.uniform(math.log(lo), math.log(hi)))))
    return max(mult, v // mult * mult) if v >= mult else max(1, v)
shapes = []
while len(shapes) < count:
    kind = rng.choice(["few", "few", "mid", "mid", "big", "tall"])
    if kind == "few":
        m, n = dim(1, 128, 1), dim(256, 65536, 8)
        if rng.random() < 0.3: m, n = n, m
    elif kind == "mid": m, n = dim(128, 4096, 8), dim(128, 8192, 8)
    elif kind == "big": m, n = dim(2048, 12288, 64), dim(2048, 12288, 64)
    else: m, n = dim(4096, 65536, 8), dim(64, 1024, 8)
    k = 64 * rng.choice([1, 2, 4, 8, 16, 24, 32, 48, 64, 96, 128, 224])
    if os.environ.get("AUDIT_FP8"): k *= 2                                            # (a K-tile = 128 fp8 values)
    if 2.0 * m * n * k > (3e11 if os.environ.get("AUDIT_F32") else 3e12) or 2.0 * (m * k + n * k + m * n) > (6e8 if os.environ.get("AUDIT_F32") else 1.2e9): continue
    shapes.append((m, n, k))
ALGOS = ["auto", "lp128", "lp256x128", "lp256w4", "lp256p", "lp256q", "stream64", "skinny", "lp256x192", "lp192x192", "lp256m16", "lp256qm"]
F32 = bool(os.environ.get("AUDIT_F32"))                 # f32 operands and output: the kernels that take them
if F32: ALGOS = ["auto", "f32", "lp256w4", "lp256p", "skinny", "stream64"]
FP8 = bool(os.environ.get("AUDIT_FP8"))                 # fp8 (e4m3) operands, bf16 C: the kernels that take them
if FP8: ALGOS = ["auto", "lp128", "lp256w4", "lp256p"]
TA = bool(os.environ.get("AUDIT_TA"))                   # lhs stored [K][M] (MatrixBatchLayout::MildlyPermuted { transposed: true }) x row-major rhs: lhs^T . grad_out, the weight-gradient product
if TA: ALGOS = ["auto", "lp128", "lp256w4"]
for nn in ((True,) if TA else (False, True)):
    algos = ALGOS + (["nnrows"] if nn else [])
    res = ab_algos.measure(cl, ev, shapes, algos, rounds=3, iters=10, nn=nn, f32=F32, c32=bool(os.environ.get("AUDIT_C32")), fp8=FP8, ta=TA)   # AUDIT_C32: bf16 operands, f32 C
    print(f"== rhs {'row-major [K][N]' if nn else '[N][K]'}: {len(shapes)} shapes (seed {seed})")
    for (m, n, k), r in res.items():
        us = {a: t for a, t in r["us"].items() if t == t}
        forced = [(a, t) for a, t in us.items() if a != "auto"]
        if not forced:
            print(f"{m:6d}x{n:6d}x{k:6d}: AUTO -> {r['auto']:9s} {us['auto']:8.1f} us   (no kernel takes it forced: re-laid out)", flush=True)
            continue
        best_a, best = min(forced, key=lambda x: x[1])
        ratio = us["auto"] / best
        flag = "  <-- BEHIND" if ratio > 1.10 and us["auto"] - best > 2.0 else ""
        every = "  ".join(f"{a} {t:.1f}" for a, t in forced) if os.environ.get("AUDIT_ALL_TIMES") else ""     # (fit data for tools/dev/tile_cost_model.py)
        print(f"{m:6d}x{n:6d}x{k:6d}: AUTO -> {r['auto']:9s} {us['auto']:8.1f} us   best {best_a:9s} {best:8.1f} us   x{ratio:.3f}{flag}" + (f"   | {every}" if every else ""), flush=True)
