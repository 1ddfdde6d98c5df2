#!/usr/bin/env python3
"""Dev (review of round 5, next #5a): where does the time go when config C3 is timed the reference's way -- a device sync around every
sample (crates/cubecl-common/src/benchmark.rs:234-241) -- instead of back to back?  Each sample is bracketed by the library's clock
probe (s_memtime against the 100 MHz reference, per CU) INSIDE the event pair, so the shader clock the sample actually ran at comes
out beside its duration; the idle gap in front of a sample is varied, and a sample may hold one or two launches.
usage (GPU box): python tools/dev/sync_gap_probe.py [size=8192]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from cubecl_amd import ElemType, Mi355Runtime, TensorHandle
from cubecl_amd import _native as N
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
cl = Mi355Runtime.client(); lib, ctx = cl.lib, cl.ctx
ev = bench.Events(cl)
a = TensorHandle.uniform(cl, (S, S), ElemType.BF16, bench.SEED, 100, -1.0, 1.0)
b = TensorHandle.uniform(cl, (S, S), ElemType.BF16, bench.SEED, 200, -1.0, 1.0)
c = cl.empty(S * S * 2)
d = bench.gemm_desc(N, S, S, S, N.DTYPE_BF16, N.DTYPE_BF16, trans_b=1)
alg = C.c_int32(); lib.mi355_gemm_select(ctx, C.byref(d), C.byref(alg))
pa, pb, pc = C.c_void_p(a.device_ptr()), C.c_void_p(b.device_ptr()), C.c_void_p(c.device_ptr())
gemm = lambda: cl._s.check(lib.mi355_gemm(ctx, None, C.byref(d), pa, pb, pc))
clk = cl.empty(2 * 8192)
lib.mi355_memset(ctx, None, C.c_void_p(clk.device_ptr()), 0, 2 * 8192)
p0, p1 = C.c_void_p(clk.device_ptr()), C.c_void_p(clk.device_ptr() + 8192)
flop = 2.0 * S ** 3


def ghz():
    tk = np.frombuffer(cl.read_one(clk), dtype=np.uint64).reshape(2, 512, 2).astype(np.float64)
    ok = (tk[0, :, 1] > 0) & (tk[1, :, 1] > tk[0, :, 1]) & (tk[1, :, 0] > tk[0, :, 0])
    if not ok.any():
        return float("nan"), float("nan")
    return (float(np.median((tk[1, ok, 0] - tk[0, ok, 0]) / (tk[1, ok, 1] - tk[0, ok, 1]) * 0.1)),
            float(np.median(tk[1, ok, 1] - tk[0, ok, 1]) * 0.01))          # GHz, microseconds between the two probes (100 MHz ticks)


print(f"C3 {S}^3 bf16, algo {alg.value}; peak 2500 TFLOP/s at 2.4 GHz", flush=True)
for _ in range(60):
    gemm()
cl.sync()
lib.mi355_probe_clock(ctx, None, p0); cl.sync()
# A: back to back
lib.mi355_probe_clock(ctx, None, p0)
ms = bench.time_op(cl, ev, gemm, 30, warmup=0)
lib.mi355_probe_clock(ctx, None, p1); cl.sync()
g, _ = ghz()
print(f"A  back to back, 30 launches between one event pair : {ms * 1e3:8.1f} us / launch  {flop / ms / 1e9:7.1f} TFLOP/s  frac {flop / ms / 1e9 / 2500:.4f}  clock {g:.3f} GHz  at-clock {flop / ms / 1e9 / (2500 * g / 2.4):.4f}", flush=True)


def sampled(launches, gap_s, samples=15, label=""):
    out, clocks, spans = [], [], []
    for _ in range(5):
        gemm()
    cl.sync()
    for _ in range(samples):
        if gap_s:
            time.sleep(gap_s)
        ev.start()
        lib.mi355_probe_clock(ctx, None, p0)
        for _ in range(launches):
            gemm()
        lib.mi355_probe_clock(ctx, None, p1)
        t = ev.stop_ms()                     # records the stop event and waits for it: the sync of the protocol
        out.append(t / launches)
        g, span = ghz()
        clocks.append(g); spans.append(span / launches)
    med = sorted(out)[len(out) // 2]
    gm = float(np.nanmedian(clocks)); sp = float(np.nanmedian(spans))
    print(f"{label:2s} per sample: {launches} launch(es), idle gap {gap_s * 1e3:6.1f} ms in front : {med * 1e3:8.1f} us / launch (events)  {sp:8.1f} us / launch (between the probes)  "
          f"{flop / med / 1e9:7.1f} TFLOP/s  frac {flop / med / 1e9 / 2500:.4f}  clock {gm:.3f} GHz  at-clock {flop / med / 1e9 / (2500 * gm / 2.4):.4f}   "
          f"min {min(out) * 1e3:.1f} max {max(out) * 1e3:.1f}", flush=True)


sampled(1, 0.0, label="B")
sampled(2, 0.0, label="C")
sampled(4, 0.0, label="C4")
sampled(1, 0.0002, label="D1")
sampled(1, 0.002, label="D2")
sampled(1, 0.02, label="D3")
sampled(1, 0.0, label="B'")
# E: the same protocol, but the NEXT sample's launch is queued before the host waits for this one (the device never idles)
out = []
evs = [bench.Events(cl) for _ in range(2)]
for _ in range(5):
    gemm()
cl.sync()
prev = None
for i in range(16):
    e = evs[i % 2]
    e.start(); gemm()
    cl._s.check(lib.mi355_event_record(ctx, e.b, None))
    if prev is not None:
        cl._s.check(lib.mi355_event_sync(ctx, prev.b))
        msv = C.c_float(); cl._s.check(lib.mi355_event_elapsed_ms(ctx, prev.a, prev.b, C.byref(msv))); out.append(msv.value)
    prev = e
out.sort()
med = out[len(out) // 2]
print(f"E  one event pair per launch, next launch queued before the wait (no idle device)  : {med * 1e3:8.1f} us / launch  {flop / med / 1e9:7.1f} TFLOP/s  frac {flop / med / 1e9 / 2500:.4f}", flush=True)
