#!/usr/bin/env python3
"""Fits the dispatcher's cost table of the tile kernels (gemm.cpp `tile_costs`) to the interleaved A/B tables under profiles/ and
reports how well the table reproduces them.  No GPU needed: the inputs are the committed `tools/ab_algos.py` tables.

Model (per kernel k, 16-bit [N][K] operands, cold):
    rounds = whole + F(left / slots),  slots = CUs x co-resident workgroups,  F(x) = 0 if x == 0 else F0 + (1 - F0) x
    T_k    = max( rounds x (nk x c_k + f_k),      the workgroups' own pace: c_k us per K-tile, f_k us of prologue + epilogue + launch
                  padded FLOPs / P_k )            the chip's pace at its power limit on random operands: P_k TFLOP/s
AUTO takes the kernel with the smallest T_k among those that support the descriptor.

The row-major [K][N] rhs has its own rows (--nn: fitted on profiles/r05_tile_nn_ab.txt): the 128-row kernels stage it through a
different image and run it slower, the 4-wave kernel's transposing reads cost it next to nothing.

usage: python tools/dev/tile_cost_model.py [--nn] [--emit]      (--emit prints the C++ table)"""
import glob
import os
import re
import sys

import numpy as np
from scipy.optimize import least_squares

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
TILES = {"lp128": (128, 128, 1),        # (two co-resident workgroups share one CU's L2 -> LDS delivery: one slot per CU, two tiles deep)
         "lp256x128": (256, 128, 1), "lp256w4": (256, 256, 1), "lp256x192": (256, 192, 1), "lp192x192": (192, 192, 1),
         "lp256m16": (256, 256, 1)}
CUS, F0 = 256, 0.7
FILES = ["r03_tile_256x128_sweep.txt", "r04_band_129_200_tiles_ab.txt", "r05_tile_256x192_ab.txt", "r05_tile_192x192_ab.txt", "r05_m16_ab.txt",
         "r05_select_audit_cost_table_v1.txt"]
if "--nn" in sys.argv:
    FILES = ["r05_tile_nn_ab.txt", "r06_long_k_tiles_nn_ab.txt"]          # (r05_tile_nn_ab2.txt is the held-out check: AUTO column = this table's choice; round 6: nine shapes past K = 8192)


def parse():
    rows = {}                                   # (m, n, k) -> {kernel: [us, ...]}
    for name in FILES:
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        cols = None
        for line in open(path):
            if line.strip().startswith("shape"):
                cols = line.split()[2:]
                continue
            au = re.match(r"(\d+)x(\d+)x(\d+): AUTO -> .*\| (.*)", line)          # tests/test_gpu_select_audit.py's table: "... | lp128 175.5  lp256w4 103.9 ..."
            if au:
                shape = tuple(int(au.group(i)) for i in (1, 2, 3))
                if min(shape[0], shape[1]) > 512:
                    for kern, t in re.findall(r"(\w+) ([\d.]+)", au.group(4)):
                        if kern in TILES:
                            rows.setdefault(shape, {}).setdefault(kern, []).append(float(t))
                continue
            m = re.match(r"\s*(\d+)x(\d+)\s*x(\d+)\s+(\S+)\s+(.*)", line)
            if not m or cols is None:
                continue
            shape = tuple(int(m.group(i)) for i in (1, 2, 3))
            times = re.findall(r"([\d.]+)us", m.group(5))
            for kern, t in zip(cols, times):
                if kern in TILES:
                    rows.setdefault(shape, {}).setdefault(kern, []).append(float(t))
    return {s: {k: float(np.median(v)) for k, v in d.items()} for s, d in rows.items()}


def rounds_of(tiles, slots):
    whole, left = divmod(tiles, slots)
    return whole + (F0 + (1 - F0) * left / slots if left else 0.0)


def predict(kern, shape, c, f, p):
    m, n, k = shape
    tm, tn, per_cu = TILES[kern]
    tiles = -(-m // tm) * -(-n // tn)
    nk = k // 64
    own = rounds_of(tiles, CUS * per_cu) * (nk * c + f)
    chip = 2.0 * tiles * tm * tn * k / (p * 1e6)       # us
    return max(own, chip)


def main():
    data = parse()
    params = {}
    for kern in TILES:
        pts = [(s, d[kern]) for s, d in data.items() if kern in d and s[2] >= 512]
        if len(pts) < 4:
            continue

        def resid(x):
            return [np.log(predict(kern, s, x[0], x[1], x[2]) / t) for s, t in pts]
        best = None
        for p0 in (800.0, 1100.0, 1400.0):
            r = least_squares(resid, x0=[0.8, 6.0, p0], bounds=([0.05, 0.0, 300.0], [5.0, 40.0, 2600.0]))
            if best is None or r.cost < best.cost:
                best = r
        params[kern] = best.x
        err = np.abs(np.exp(resid(best.x)) - 1.0)
        print(f"{kern:10s} c = {best.x[0]:.3f} us per K-tile, f = {best.x[1]:5.2f} us, P = {best.x[2]:6.0f} TFLOP/s   {len(pts):3d} points, "
              f"median |err| {100 * np.median(err):.1f} %, max {100 * err.max():.1f} %")
    # how good are the table's decisions?  regret = time of the model's choice / best measured time, over shapes with >= 2 kernels measured
    regrets, wrong = [], []
    for s, d in sorted(data.items()):
        cand = [k for k in d if k in params]
        if len(cand) < 2 or s[2] < 512:
            continue
        choice = min(cand, key=lambda k: predict(k, s, *params[k]))
        best = min(cand, key=lambda k: d[k])
        regrets.append(d[choice] / d[best])
        if d[choice] / d[best] > 1.05:
            wrong.append((s, choice, best, d[choice] / d[best]))
    regrets = np.array(regrets)
    print(f"decisions over {len(regrets)} measured shapes: mean regret {100 * (regrets.mean() - 1):.2f} %, worst {100 * (regrets.max() - 1):.1f} %, "
          f"{int((regrets > 1.05).sum())} shapes more than 5 % behind, {int((regrets > 1.10).sum())} more than 10 %")
    for s, choice, best, r in wrong:
        print(f"    {s[0]} x {s[1]} x {s[2]}: table takes {choice}, measured best {best} (+{100 * (r - 1):.0f} %)")
    if "--emit" in sys.argv:
        ids = {"lp128": "MI355_GEMM_ALGO_LP_128", "lp256x128": "MI355_GEMM_ALGO_LP_256X128", "lp256w4": "MI355_GEMM_ALGO_LP_256W4",
               "lp256x192": "MI355_GEMM_ALGO_LP_256X192", "lp192x192": "MI355_GEMM_ALGO_LP_192X192", "lp256m16": "MI355_GEMM_ALGO_LP_256M16"}
        print("// generated by tools/dev/tile_cost_model.py --emit from profiles/" + ", ".join(FILES))
        for kern, x in params.items():
            tm, tn, per_cu = TILES[kern]
            print(f"    {{{ids[kern]}, {tm}, {tn}, {per_cu}, {x[0]:.3f}, {x[1]:.2f}, {x[2]:.0f}.0}},")


if __name__ == "__main__":
    main()
