#!/usr/bin/env python3
"""Dev: config C4 at the size BASELINE defines it -- the 128 MiB slice one of 8 GPUs owns -- on the one GPU there is.
The passes walk the eight slices of ONE 1 GiB array in rotation (every slice is cold again when its turn comes: 1 GiB
is four times the 256 MiB infinity cache), back to back between one event pair and per sample.
usage (GPU box): python tools/dev/shard_probe.py [shards=8] [total_elements=2^28]
  MI355_REDUCE_WG_PER_CU=n   workgroups per CU of the array-wide kernel (library default otherwise)
  with a -DRED_TRACE build (tools/dev/build_variants.sh redtrace "-DRED_TRACE" reduce.hip; MI355CUBE_LIB=...) the phases
  of one pass per workgroup are printed too."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from cubecl_amd import ElemType, Mi355Runtime, TensorHandle
cl = Mi355Runtime.client(); lib, ctx = cl.lib, cl.ctx
shards = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 28
ns = n // shards
x = TensorHandle.uniform(cl, (n,), ElemType.F32, 1, 300, 0.0, 1.0)
ws = cl.empty(1 << 17); outs = cl.empty(64)
ea, eb = C.c_void_p(), C.c_void_p(); lib.mi355_event_create(ctx, C.byref(ea)); lib.mi355_event_create(ctx, C.byref(eb))
p_ws = C.c_void_p(ws.device_ptr())
p_sum, p_val, p_idx = (C.c_void_p(outs.device_ptr() + o) for o in (0, 8, 16))
slices = [C.c_void_p(x.device_ptr() + 4 * ns * i) for i in range(shards)]
fns = {"sum": lambda p: lib.mi355_reduce_sum_f32(ctx, None, p, ns, p_sum, p_ws, ws.size),
       "argmax": lambda p: lib.mi355_argmax_f32(ctx, None, p, ns, p_val, p_idx, p_ws, ws.size),
       "fused": lambda p: lib.mi355_sum_argmax_f32(ctx, None, p, ns, p_sum, p_val, p_idx, p_ws, ws.size)}


def elapsed():
    ms = C.c_float(); lib.mi355_event_elapsed_ms(ctx, ea, eb, C.byref(ms)); return ms.value * 1e3


print(f"{shards} slices of {ns * 4 / 2**20:.0f} MiB; MI355_REDUCE_WG_PER_CU={os.environ.get('MI355_REDUCE_WG_PER_CU', 'default')}", flush=True)
for name, fn in fns.items():
    for r in range(3 * shards): fn(slices[r % shards])
    cl.sync()
    b2b = []
    for rep in range(5):
        lib.mi355_event_record(ctx, ea, None)
        for r in range(4 * shards): fn(slices[r % shards])
        lib.mi355_event_record(ctx, eb, None); lib.mi355_event_sync(ctx, eb)
        b2b.append(elapsed() / (4 * shards))
    per = []
    for r in range(4 * shards):
        lib.mi355_event_record(ctx, ea, None); fn(slices[r % shards]); lib.mi355_event_record(ctx, eb, None); lib.mi355_event_sync(ctx, eb)
        per.append(elapsed())
    b2b.sort(); per.sort()
    print(f"{name:7s} back to back median {b2b[2]:6.2f} us min {b2b[0]:6.2f} us ({ns * 4 / b2b[2] / 1e3:7.1f} GB/s = {ns * 4 / b2b[2] / 8e6:.3f} of 8 TB/s)   "
          f"per sample median {per[len(per) // 2]:6.2f} min {per[0]:6.2f} us", flush=True)

if hasattr(lib, "mi355_dev_red_trace"):
    buf = np.zeros(4096 * 8, dtype=np.uint64)
    prev = None
    for name in ("sum", "sum", "fused"):
        for i in range(shards + 3):
            cl.sync()
            lib.mi355_dev_red_trace(buf.ctypes.data_as(C.c_void_p), 1)
            fns[name](slices[i % shards])
            cl.sync()
        lib.mi355_dev_red_trace(buf.ctypes.data_as(C.c_void_p), 0)
        t = buf.reshape(4096, 8).astype(np.float64)
        t = t[t[:, 0] > 0]
        t0 = t[:, 0].min()
        names = ["entry", "whole rounds done", "dealt rows done", "record stored", "ticket back", "fold done (last workgroup)"]
        print(f"{name}: {len(t)} workgroups; s_memrealtime ticks (100 MHz: 1 tick = 10 ns), relative to the first workgroup's entry")
        for j, nm in enumerate(names):
            col = t[:, j][t[:, j] > 0]
            if len(col):
                rel = col - t0
                print(f"  {nm:28s} n={len(col):4d}  min {rel.min():6.0f}  p10 {np.percentile(rel, 10):6.0f}  median {np.median(rel):6.0f}  p90 {np.percentile(rel, 90):6.0f}  max {rel.max():6.0f}")
        done = t[:, 2] - t0
        print("  streaming done per XCD (workgroup index % 8): " + "  ".join(f"{x}: {np.median(done[x::8]):5.0f}/{done[x::8].max():5.0f}" for x in range(8)) + "   (median/max)")
        ent = t[:, 0] - t0
        dur = t[:, 2] - t[:, 0]
        print("  entry per XCD (median/max):              " + "  ".join(f"{x}: {np.median(ent[x::8]):5.0f}/{ent[x::8].max():5.0f}" for x in range(8)))
        print("  streaming duration per XCD (done - entry): " + "  ".join(f"{x}: {np.median(dur[x::8]):5.0f}/{dur[x::8].max():5.0f}" for x in range(8)))
        order = np.arange(len(t)) // 8
        print("  by dispatch order within the XCD (index / 8), quarters: entry " + "  ".join(f"{np.median(ent[(order >= 8 * i) & (order < 8 * i + 8)]):5.0f}" for i in range(4))
              + "   duration " + "  ".join(f"{np.median(dur[(order >= 8 * i) & (order < 8 * i + 8)]):5.0f}" for i in range(4))
              + "   done " + "  ".join(f"{np.median(done[(order >= 8 * i) & (order < 8 * i + 8)]):5.0f}" for i in range(4)))
        q = len(done) // 4
        print("  streaming done by quarter of the grid: " + "  ".join(f"{np.median(done[i * q:(i + 1) * q]):5.0f}" for i in range(4)))
        if prev is not None and len(prev) == len(done):
            print(f"  correlation with the previous traced pass: {np.corrcoef(prev, done)[0, 1]:.3f}")
        prev = done
