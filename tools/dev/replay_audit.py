#!/usr/bin/env python3
"""Replays a seeded audit's timed cases (AUDIT_ALL_TIMES=1 tools/dev/random_audit.py: every kernel's time for every shape; AUDIT_F32=1 f32 operands, AUDIT_C32=1 bf16
operands with an f32 C) against the dispatcher as built now:
`mi355_gemm_select` is a host function of the descriptor alone (ctx may be NULL), so what AUTO would pick today -- and how far behind the fastest timed kernel that pick is --
can be counted without a GPU.  usage: python tools/dev/replay_audit.py [--f32 | --c32 | --bf16 | --fp8 | --ta] [files ...]   (default: --f32 on the two f32 files)"""
import os, sys, re, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cubecl_amd import _native as N
lib = N.load()
NAME = {N.GEMM_ALGO_F32_MFMA: "f32", N.GEMM_ALGO_LP_256W4: "lp256w4", N.GEMM_ALGO_LP_256P: "lp256p", N.GEMM_ALGO_SKINNY: "skinny", N.GEMM_ALGO_STREAM64: "stream64",
        N.GEMM_ALGO_NNROWS: "nnrows", N.GEMM_ALGO_GENERIC: "generic", N.GEMM_ALGO_LP_128: "lp128", N.GEMM_ALGO_LP_256X128: "lp256x128", N.GEMM_ALGO_LP_256Q: "lp256q",
        N.GEMM_ALGO_LP_256X192: "lp256x192", N.GEMM_ALGO_LP_192X192: "lp192x192", N.GEMM_ALGO_LP_256M16: "lp256m16", N.GEMM_ALGO_LP_256QM: "lp256qm"}
args = [x for x in sys.argv[1:] if not x.startswith("--")]
mode = ([x for x in sys.argv[1:] if x.startswith("--")] or ["--f32"])[0]
DT_AB, DT_C = {"--f32": (N.DTYPE_F32, N.DTYPE_F32), "--c32": (N.DTYPE_BF16, N.DTYPE_F32), "--bf16": (N.DTYPE_BF16, N.DTYPE_BF16), "--fp8": (N.DTYPE_F8E4M3, N.DTYPE_BF16), "--ta": (N.DTYPE_BF16, N.DTYPE_BF16)}[mode]
TA = mode == "--ta"                                     # lhs stored [K][M]
files = args or ["profiles/r06_audit_times_f32_fit.txt", "profiles/r06_audit_times_f32_held_out.txt"]
for path in files:
    nn, cases, behind, untimed, regret = 0, 0, [], 0, 0.0
    for line in open(os.path.join(ROOT, path)):
        if line.startswith("== rhs"): nn = 1 if "row-major" in line else 0; continue
        mt = re.match(r"\s*(\d+)x\s*(\d+)x\s*(\d+): AUTO -> (\S+)\s+([\d.]+) us.*\| (.*)", line)
        if not mt: continue
        m, n, k = (int(mt.group(i)) for i in (1, 2, 3))
        us = {a: float(t) for a, t in re.findall(r"(\w+) ([\d.]+)", mt.group(6))}
        # the forced 256 x 256 launch is the plain one; AUTO's is split where the launcher plans a tail split: where AUTO chose the square tile when the file was
        # taken, its time is what a pick of the square tile gets today
        if mt.group(4) == "lp256w4" and "lp256w4" in us: us["lp256w4"] = min(us["lp256w4"], float(mt.group(5)))
        d = N.GemmDesc(m=m, n=n, k=k, batch=1, lda=m if TA else k, ldb=n if nn else k, ldc=n, stride_a=m * k, stride_b=n * k, stride_c=m * n, dtype_ab=DT_AB, dtype_c=DT_C,
                       trans_a=1 if TA else 0, trans_b=0 if nn else 1, algo=N.GEMM_ALGO_AUTO)
        algo = C.c_int32(-1)
        assert lib.mi355_gemm_select(None, C.byref(d), C.byref(algo)) == N.OK
        pick = NAME.get(algo.value, str(algo.value))
        if pick not in us: untimed += 1; continue
        cases += 1
        best = min(us, key=us.get)
        regret += us[pick] / us[best] - 1.0
        if us[pick] > 1.1 * us[best] and us[pick] - us[best] > 2.0: behind.append((nn, m, n, k, pick, us[pick], best, us[best]))
    print(f"{path}: {cases} cases replayed ({untimed} picks without a time), {len(behind)} more than 10 % and 2 us behind, mean regret {100 * regret / max(cases, 1):.2f} %")
    for nn, m, n, k, pick, t, best, tb in behind:
        print(f"  {'row-major' if nn else '[N][K]   '} {m:6d}x{n:6d}x{k:6d}: AUTO -> {pick:9s} {t:8.1f} us   best {best:9s} {tb:8.1f} us   x{t / tb:.3f}")
