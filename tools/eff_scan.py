#!/usr/bin/env python3
"""Efficiency scan of AUTO over a random grid of bf16 shapes (GPU box, cold operands): time against a crude ideal
= max(FLOP / 1.5 PFLOP/s, bytes / 5 TB/s) + 2.5 us of launch; prints the shapes furthest from it.  A tool for finding
dispatcher / launcher decisions that are plainly wrong (a forced-kernel A/B cannot see a bad decision INSIDE a kernel's launcher).
usage: tools/eff_scan.py [seed] [count] [nn]      (nn: rhs row-major [K][N])"""
import random
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(Path(__file__).resolve().parent))
import ab_algos  # noqa: E402
import bench  # noqa: E402
from cubecl_amd import Mi355Runtime  # noqa: E402

rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
DIMS = [1, 2, 4, 8, 16, 32, 48, 64, 96, 128, 192, 256, 384, 512, 768, 1024, 1536, 2048, 3072, 4096, 6144, 8192, 14336, 28672]
KS = [64, 128, 256, 512, 1024, 2048, 4096, 8192, 14336, 16384]
shapes = set()
while len(shapes) < count:
    m, n, k = rng.choice(DIMS), rng.choice(DIMS), rng.choice(KS)
    if m * n * 2 > (1 << 28) or (m * k + n * k) * 2 > (1 << 29) or m * n * k < (1 << 24):
        continue
    shapes.add((m, n, k))
client = Mi355Runtime.client()
ev = bench.Events(client)
res = ab_algos.measure(client, ev, sorted(shapes), ["auto"], rounds=3, iters=10, nn=len(sys.argv) > 3 and sys.argv[3] == "nn")
rows = []
for (m, n, k), r in res.items():
    us = r["us"]["auto"]
    ideal = max(2.0 * m * n * k / 1.5e15, 2.0 * (m * k + n * k + m * n) / 5.0e12) * 1e6 + 2.5
    rows.append((ideal / us, m, n, k, r["auto"], us, ideal))
rows.sort()
for eff, m, n, k, algo, us, ideal in rows:
    print(f"{eff:5.2f}  {m:>6d}x{n:<6d}x{k:<6d} {algo:>10s} {us:9.1f} us   ideal {ideal:8.1f}")
