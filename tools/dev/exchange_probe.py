#!/usr/bin/env python3
"""Dev: what the exchange step of config C4 costs on ONE rank of the real RCCL (launch, fences, combine kernel -- no wire):
one all-gather + mi355_sum_argmax_combine_f32 against the reference's shape (all_reduce + all-gather + combine), each alone
and behind the fused 128 MiB shard pass.  MI355_COMM_INLINE_BYTES=0 puts every collective on the communication stream
between the two event fences (the form of rounds 1-4); the default queues messages <= 4 KiB in the compute stream's order.
usage (GPU box): [MI355_COMM_INLINE_BYTES=0] python tools/dev/exchange_probe.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from cubecl_amd import DeviceId, ElemType, Mi355Runtime, TensorHandle, ops, sharded
cl = Mi355Runtime.client(); lib, ctx = cl.lib, cl.ctx
n = 1 << 28; SH = 8; ns = n // SH
x = TensorHandle.uniform(cl, (n,), ElemType.F32, 1, 300, 0.0, 1.0)
ws = cl.empty(1 << 17); outs = cl.empty(64)
ea, eb = C.c_void_p(), C.c_void_p(); lib.mi355_event_create(ctx, C.byref(ea)); lib.mi355_event_create(ctx, C.byref(eb))
p_ws = C.c_void_p(ws.device_ptr())
p_val, p_sum, p_idx = (C.c_void_p(outs.device_ptr() + o) for o in (0, 4, 8))
ids = [DeviceId(0, 0)]
cl.comm_init(ids, bytes(cl.comm_unique_id()), rank=0)
ex = sharded.RcclExchange(cl, ids, 0)
rec = outs.offset_end_by(48)
g_sum, g_val, g_idx = outs.offset_start_by(32).offset_end_by(28), outs.offset_start_by(36).offset_end_by(24), outs.offset_start_by(40).offset_end_by(16)
turn = [0]


def shard_pass():
    turn[0] = (turn[0] + 1) % SH
    cl._s.check(lib.mi355_sum_argmax_f32(ctx, None, C.c_void_p(x.device_ptr() + 4 * ns * turn[0]), ns, p_sum, p_val, p_idx, p_ws, ws.size))


def timed(fn, iters=32, reps=5):
    for _ in range(8): fn()
    cl.sync(); out = []
    for _ in range(reps):
        lib.mi355_event_record(ctx, ea, None)
        for _ in range(iters): fn()
        lib.mi355_event_record(ctx, eb, None); lib.mi355_event_sync(ctx, eb)
        ms = C.c_float(); lib.mi355_event_elapsed_ms(ctx, ea, eb, C.byref(ms)); out.append(ms.value * 1e3 / iters)
    out.sort()
    return out[len(out) // 2], out[0]


gathered = ex._buf.offset_start_by(64)
cases = {
    "shard pass alone (fused, 128 MiB, cold)": shard_pass,
    "combine kernel alone": lambda: ops.sum_argmax_combine(cl, gathered, 1, [0], g_sum, g_val, g_idx),
    "all-gather + sync_collective alone": lambda: (cl.all_gather(rec, gathered, ElemType.U64, ids), cl.sync_collective()),
    "exchange, one collective": lambda: ex.exchange_on_device(rec, [0], g_sum, g_val, g_idx),
    "exchange, all_reduce + all-gather": lambda: ex.exchange_on_device(rec, [0], g_sum, g_val, g_idx, mode="all_reduce"),
    "shard pass + exchange (one collective)": lambda: (shard_pass(), ex.exchange_on_device(rec, [0], g_sum, g_val, g_idx)),
    "shard pass + exchange (two collectives)": lambda: (shard_pass(), ex.exchange_on_device(rec, [0], g_sum, g_val, g_idx, mode="all_reduce")),
}
print(f"MI355_COMM_INLINE_BYTES={os.environ.get('MI355_COMM_INLINE_BYTES', 'default (4096)')}; back to back on the compute stream, us per call (median / min of 5 x 32)")
for name, fn in cases.items():
    med, best = timed(fn)
    print(f"  {name:44s} {med:7.2f} / {best:7.2f}", flush=True)
got = np.frombuffer(cl.read_one(outs), dtype=np.uint8)
print("  combine of one record reproduces the local pass:", bytes(got[0:4]) == bytes(got[36:40]) and bytes(got[4:8]) == bytes(got[32:36]) and bytes(got[8:16]) == bytes(got[40:48]))
