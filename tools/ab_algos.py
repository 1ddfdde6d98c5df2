#!/usr/bin/env python3
"""Interleaved A/B of GEMM kernels over a list of shapes (runs on the GPU box).

usage: tools/ab_algos.py [--rounds R] [--nn] [--algos auto,lp128,lp256x128,...] MxNxK [MxNxK ...]

Every round times every algorithm once (20 launches between one event pair, rotating through enough operand sets that no
launch finds its operands in the Infinity Cache), algorithms interleaved so that clock / thermal drift hits all of them alike;
the table prints the median over rounds in microseconds and TFLOP/s, and AUTO's choice."""
import argparse
import ctypes as C
import statistics
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402
from cubecl_amd import ElemType, Mi355Runtime, TensorHandle  # noqa: E402
from cubecl_amd import _native as N  # noqa: E402

VERBOSE = False
NAMES = {"auto": 0, "generic": 1, "f32": 2, "lp128": 3, "lp256": 4, "lp256w4": 5, "lp256p": 6, "lp256q": 7, "skinny": 8, "stream64": 9,
         "lp256x128": 10, "nnrows": 11, "lp256x192": 12, "lp192x192": 13, "lp256m16": 14, "lp256qm": 15}
BY_ID = {v: k for k, v in NAMES.items()}


def measure(client, ev, shapes, algos, rounds=5, nn=False, iters=20, cold=True, f32=False, c32=False, fp8=False, ta=False):
    lib, ctx = client.lib, client.ctx
    et, dt, esz = (ElemType.F32, N.DTYPE_F32, 4) if f32 else (ElemType.BF16, N.DTYPE_BF16, 2)
    if fp8: et, dt, esz = ElemType.F8E4M3, N.DTYPE_F8E4M3, 1          # fp8 operands; C bf16 (or f32 with c32)
    dtc, csz = (N.DTYPE_F32, 4) if c32 else (N.DTYPE_BF16, 2) if fp8 else (dt, esz)          # c32: 16-bit operands, f32 C (the cmma tests' accumulator type as the output)
    out = {}
    for (m, n, k) in shapes:
        fp = esz * (m * k + n * k) + csz * m * n
        nsets = max(1, min(8, -(-(768 << 20) // fp))) if cold else 1
        sets = [(TensorHandle.uniform(client, (m, k), et, 1, 2 * i + 1, -1.0, 1.0),
                 TensorHandle.uniform(client, (n, k), et, 1, 2 * i + 2, -1.0, 1.0), client.empty(m * n * csz)) for i in range(nsets)]
        times = {a: [] for a in algos}
        sel = C.c_int32()
        d0 = N.GemmDesc(m=m, n=n, k=k, batch=1, lda=(m if ta else k), trans_a=1 if ta else 0, ldb=(n if nn else k), ldc=n, dtype_ab=dt, dtype_c=dtc, trans_b=0 if nn else 1, algo=0)
        lib.mi355_gemm_select(ctx, C.byref(d0), C.byref(sel))
        turn = [0]
        for _ in range(rounds):
            for a in algos:
                d = N.GemmDesc(m=m, n=n, k=k, batch=1, lda=(m if ta else k), trans_a=1 if ta else 0, ldb=(n if nn else k), ldc=n, dtype_ab=dt, dtype_c=dtc,
                               trans_b=0 if nn else 1, algo=NAMES[a])

                def call():
                    sa, sb, sc = sets[turn[0] % nsets]
                    turn[0] += 1
                    rc = lib.mi355_gemm(ctx, None, C.byref(d), sa.device_ptr(), sb.device_ptr(), sc.device_ptr())
                    if rc != N.OK:
                        raise RuntimeError(rc)
                if times[a] and times[a][-1] != times[a][-1]:
                    continue            # refused in the first round: not asked again (round 6: a refused launch in front of AUTO's turn cost AUTO up to a third on
                                        # output-bound shapes in tools/dev/batched_audit.py -- the same kernel forced, two turns later, ran at its usual time)
                if VERBOSE:
                    print(f"[ab] {m}x{n}x{k} {a} nn={nn} sets={nsets}", file=sys.stderr, flush=True)
                try:
                    times[a].append(bench.time_op(client, ev, call, iters, warmup=3) * 1e3)
                except RuntimeError:
                    times[a].append(float("nan"))
                    client.flush_errors() if hasattr(client, "flush_errors") else None
        out[(m, n, k)] = {"auto": BY_ID.get(sel.value, str(sel.value)),
                          "us": {a: statistics.median(v) for a, v in times.items()}}   # (a refused algorithm: [nan])
        del sets
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--nn", action="store_true")
    ap.add_argument("--f32", action="store_true", help="f32 operands and output instead of bf16")
    ap.add_argument("--verbose", action="store_true", help="name every measurement on stderr before it starts")
    ap.add_argument("--warm", action="store_true", help="one operand set (Infinity-Cache-warm operands)")
    ap.add_argument("--algos", default="auto,lp128,lp256x128,lp256w4")
    ap.add_argument("shapes", nargs="+")
    args = ap.parse_args()
    global VERBOSE
    VERBOSE = args.verbose
    client = Mi355Runtime.client()
    ev = bench.Events(client)
    shapes = [tuple(int(x) for x in s.split("x")) for s in args.shapes]
    algos = args.algos.split(",")
    res = measure(client, ev, shapes, algos, args.rounds, args.nn, cold=not args.warm, f32=args.f32)
    print(f"{'shape':>20s} {'AUTO':>10s} " + " ".join(f"{a:>16s}" for a in algos))
    for (m, n, k), r in res.items():
        cells = []
        for a in algos:
            us = r["us"][a]
            cells.append(f"{us:8.1f}us {2.0 * m * n * k / us / 1e6:6.0f}T" if us == us else f"{'n/a':>16s}")
        print(f"{m:>6d}x{n:<6d}x{k:<6d} {r['auto']:>10s} " + " ".join(cells))


if __name__ == "__main__":
    main()
