#!/usr/bin/env python3
"""Dev (GPU box): AUTO against every forced kernel on BATCHED bf16 GEMMs (contiguous strided batches, the layout of BASELINE config C5):
attention-shaped products (heads x [seq x seq x 128] and [seq x 128 x seq]), expert-shaped ones (8 ... 64 x [tokens x ffn x hidden]), cubes;
both rhs layouts, cold operands.  tools/dev/random_audit.py and tools/ab_algos.py only time batch = 1.  usage: tools/dev/batched_audit.py [BxMxNxK ...]"""
import ctypes as C, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ab_algos, bench
from cubecl_amd import ElemType, Mi355Runtime, TensorHandle
from cubecl_amd import _native as N
cl = Mi355Runtime.client(); ev = bench.Events(cl); lib, ctx = cl.lib, cl.ctx
SHAPES = []
for heads in (8, 32, 64):
    for seq in (512, 1024, 2048, 4096):
        SHAPES += [(heads, seq, seq, 128), (heads, seq, 128, seq)]
for e, t in ((8, 512), (8, 2048), (16, 1024), (64, 256), (64, 64), (8, 16)):
    SHAPES += [(e, t, 14336, 4096), (e, t, 4096, 14336)]
SHAPES += [(b, s, s, s) for b, s in ((4, 4096), (16, 2048), (64, 1024), (64, 2048), (256, 512), (512, 256), (24, 1536), (10, 2304), (3, 3072))]
if len(sys.argv) > 1:                        # explicit list: BxMxNxK ...
    SHAPES = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
SHAPES = [s for s in SHAPES if 2.0 * s[0] * s[1] * s[2] * s[3] <= 6e12 and 2.0 * s[0] * (s[1] * s[3] + s[2] * s[3] + s[1] * s[2]) <= (2.0e9 if os.environ.get("AUDIT_C32") else 3.0e9)]
ALGOS = ["auto", "lp128", "lp256x128", "lp256w4", "lp256p", "lp256q", "stream64", "skinny", "lp256x192", "lp192x192", "lp256m16", "lp256qm"]
if os.environ.get("AUDIT_ORDER"):            # dev: another measurement order (is a difference the kernel or its place in the round?)
    ALGOS = os.environ["AUDIT_ORDER"].split(",")
C32 = bool(os.environ.get("AUDIT_C32"))          # bf16 operands, f32 C
CSZ, DTC = (4, N.DTYPE_F32) if C32 else (2, N.DTYPE_BF16)
TA = bool(os.environ.get("AUDIT_TA"))            # lhs stored [K][M] per matrix x row-major rhs (the weight-gradient / attention-backward products)
if TA: ALGOS = ["auto", "lp128", "lp256w4"]
behind = total = 0
for nn in ((True,) if TA else (False, True)):
    print(f"== rhs {'row-major [K][N]' if nn else '[N][K]'}: {len(SHAPES)} shapes  (batch x M x N x K)")
    for (bt, m, n, k) in SHAPES:
        fp = 2 * bt * (m * k + n * k) + CSZ * bt * m * n
        nsets = max(1, min(4, -(-(768 << 20) // fp)))
        sets = [(TensorHandle.uniform(cl, (bt * m, k), ElemType.BF16, 1, 2 * i + 1, -1.0, 1.0),
                 TensorHandle.uniform(cl, (bt * n, k), ElemType.BF16, 1, 2 * i + 2, -1.0, 1.0), cl.empty(bt * m * n * CSZ)) for i in range(nsets)]
        def desc(algo):
            return N.GemmDesc(m=m, n=n, k=k, batch=bt, lda=(m if TA else k), trans_a=1 if TA else 0, ldb=(n if nn else k), ldc=n, stride_a=m * k, stride_b=n * k, stride_c=m * n,
                              dtype_ab=N.DTYPE_BF16, dtype_c=DTC, trans_b=0 if nn else 1, algo=algo)
        sel = C.c_int32(); lib.mi355_gemm_select(ctx, C.byref(desc(0)), C.byref(sel))
        times = {a: [] for a in ALGOS}; turn = [0]
        for _ in range(3):
            for a in ALGOS:
                d = desc(ab_algos.NAMES[a])
                def call():
                    sa, sb, sc = sets[turn[0] % nsets]; turn[0] += 1
                    if lib.mi355_gemm(ctx, None, C.byref(d), sa.device_ptr(), sb.device_ptr(), sc.device_ptr()) != N.OK: raise RuntimeError
                if times[a] and times[a][-1] != times[a][-1]:
                    continue                                # refused in the first round: not asked again (a refused launch queues an error on the context)
                try: times[a].append(bench.time_op(cl, ev, call, 10, warmup=2) * 1e3)
                except RuntimeError:
                    times[a].append(float("nan"))
                    if hasattr(cl, "flush_errors"): cl.flush_errors()
        us = {a: statistics.median(v) for a, v in times.items() if v}; us = {a: t for a, t in us.items() if t == t}
        forced = [(a, t) for a, t in us.items() if a != "auto"]
        del sets
        if not forced or "auto" not in us:
            print(f"{bt:4d} x {m:5d}x{n:5d}x{k:5d}: AUTO -> {ab_algos.BY_ID.get(sel.value)} (nothing to compare)"); continue
        ba, best = min(forced, key=lambda x: x[1]); ratio = us["auto"] / best
        flag = "  <-- BEHIND" if ratio > 1.10 and us["auto"] - best > 2.0 else ""
        behind += bool(flag); total += 1
        print(f"{bt:4d} x {m:5d}x{n:5d}x{k:5d}: AUTO -> {ab_algos.BY_ID.get(sel.value):9s} {us['auto']:8.1f} us {2.0 * bt * m * n * k / us['auto'] / 1e6:6.0f} TFLOP/s   best {ba:9s} {best:8.1f} us   x{ratio:.3f}{flag}" + ("   | " + "  ".join(f"{a} {t:.1f}" for a, t in forced) if os.environ.get("AUDIT_ALL_TIMES") else ""), flush=True)
print(f"{behind} of {total} more than 10 % + 2 us behind")
