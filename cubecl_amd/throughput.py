"""Measured ceilings of the device: the reference's throughput probes and what is built on them.

Mirrors crates/cubecl-runtime/src/throughput/{base,curve}.rs and crates/cubecl-std/src/throughput/base.rs:
  * `working_set_sweep`   curve.rs:35-53      powers of two from 8 KiB up to the cap
  * `MemoryCurve`         curve.rs:69-147     ascending points, `ceiling_at` interpolates linearly in log2(bytes)
  * `measure_memory_curve`   std/throughput/base.rs:46-64   one probe per working set
  * `measure_peak_throughput` :79-141          copy / read / write / compute-direct / cmma / launch
  * `roofline_bounds` + `time_limit`   :147-170, tune/bounds_generator.rs:112-160
  * `measure_peak_throughput(client, key)` / `device_throughput`   :30-44, :77-141   one probe, keyed and sampled as
    `ThroughputBenchmarker` prescribes (cubecl_amd/roofline.py), cached per device

The kernels are the library's probes (`mi355_probe_*`, cubecl_amd/csrc/probes.hip); everything here is the host
logic around them.  A working set is measured the way the reference's `MemoryProbe` does it
(std/throughput/runners/memory_probe.rs:10-43): a WINDOW of that size moves over a large buffer, a fresh window each
pass, so a small working set stays cold instead of being served from cache -- the curve describes how much a pass
moves, not what happens to be resident.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Iterable, List, Optional, Sequence

from . import _native as N
from .roofline import KernelConfig, MemoryAccess, ThroughputKey, ThroughputValue

MIN_WORKING_SET = 8 * 1024                    # curve.rs:25
DEFAULT_BUFFER_BYTES = 512 * 1024 * 1024      # throughput/base.rs:9
PROBE_TILE_BYTES = 32 * 1024                  # the probes move whole 32 KiB tiles (probes.hip PR_BLOCK * PR_UNROLL * 16)


def working_set_sweep(cap: int) -> List[int]:
    """curve.rs:35-53."""
    if cap < MIN_WORKING_SET:
        return [cap]
    sizes, b = [], MIN_WORKING_SET
    while b <= cap:
        sizes.append(b)
        b *= 2
    return sizes


def _log2(b: int) -> float:
    """curve.rs:149-161: exponent from the bit width, mantissa interpolated linearly (exact on powers of two)."""
    b = max(int(b), 1)
    e = b.bit_length() - 1
    return e + (b / (1 << e) - 1.0)


@dataclass(frozen=True)
class MemoryPoint:
    bytes: int
    bytes_per_s: float


class MemoryCurve:
    """curve.rs:69-147."""

    def __init__(self, access: MemoryAccess, points: Iterable[MemoryPoint]):
        self.access = access
        good = [p for p in points if p.bytes > 0 and p.bytes_per_s == p.bytes_per_s and 0.0 < p.bytes_per_s < float("inf")]
        good.sort(key=lambda p: p.bytes)
        out: List[MemoryPoint] = []
        for p in good:                                   # duplicated working sets keep the first point
            if not out or out[-1].bytes != p.bytes:
                out.append(p)
        self._points = out

    def points(self) -> Sequence[MemoryPoint]:
        return tuple(self._points)

    def ceiling_at(self, nbytes: int) -> Optional[float]:
        if not self._points:
            return None
        first, last = self._points[0], self._points[-1]
        if nbytes <= first.bytes:
            return first.bytes_per_s
        if nbytes >= last.bytes:
            return last.bytes_per_s
        above = next(i for i, p in enumerate(self._points) if p.bytes > nbytes)
        low, high = self._points[above - 1], self._points[above]
        w = (_log2(nbytes) - _log2(low.bytes)) / (_log2(high.bytes) - _log2(low.bytes))
        return low.bytes_per_s + w * (high.bytes_per_s - low.bytes_per_s)


# ---- measurement ------------------------------------------------------------------------------------------------------
class _Timer:
    def __init__(self, client):
        self.c = client
        self.a, self.b = C.c_void_p(), C.c_void_p()
        client._s.check(client.lib.mi355_event_create(client.ctx, C.byref(self.a)))
        client._s.check(client.lib.mi355_event_create(client.ctx, C.byref(self.b)))

    def run(self, fn, passes: int) -> float:
        """Seconds for `passes` back-to-back calls of fn(i) on the client's stream (device time)."""
        c = self.c
        c._s.check(c.lib.mi355_event_record(c.ctx, self.a, c.stream))
        for i in range(passes):
            fn(i)
        c._s.check(c.lib.mi355_event_record(c.ctx, self.b, c.stream))
        c._s.check(c.lib.mi355_event_sync(c.ctx, self.b))
        ms = C.c_float()
        c._s.check(c.lib.mi355_event_elapsed_ms(c.ctx, self.a, self.b, C.byref(ms)))
        return float(ms.value) * 1e-3

    def close(self):
        for e in (self.a, self.b):
            self.c.lib.mi355_event_destroy(self.c.ctx, e)


def measure_working_set(client, access: MemoryAccess, working_set: int, *, pool_bytes: int = DEFAULT_BUFFER_BYTES,
                        min_seconds: float = 2e-3) -> float:
    """Bytes moved per second when every pass moves `working_set` bytes (split over the buffers `access` touches)
    through a window that rotates over `pool_bytes`-sized buffers.  Passes are repeated until `min_seconds` of
    device time have been measured (the reference calibrates its iteration count the same way, benchmarker.rs)."""
    lib, ctx, st = client.lib, client.ctx, client.stream
    per_buf = max(working_set // access.buffers(), PROBE_TILE_BYTES) // PROBE_TILE_BYTES * PROBE_TILE_BYTES
    pool_bytes = max(pool_bytes // PROBE_TILE_BYTES * PROBE_TILE_BYTES, per_buf)
    windows = max(pool_bytes // per_buf, 1)
    src = client.empty(pool_bytes)
    dst = client.empty(pool_bytes) if access is MemoryAccess.Copy else None
    sink = client.empty(256)
    client._s.check(lib.mi355_memset(ctx, st, C.c_void_p(src.device_ptr()), 0, pool_bytes))
    sp, sk = src.device_ptr(), C.c_void_p(sink.device_ptr())
    dp = dst.device_ptr() if dst is not None else 0

    def one(i):
        off = (i % windows) * per_buf
        if access is MemoryAccess.Read:
            client._s.check(lib.mi355_probe_memory_read(ctx, st, C.c_void_p(sp + off), per_buf, 1, sk))
        elif access is MemoryAccess.Write:
            client._s.check(lib.mi355_probe_memory_write(ctx, st, C.c_void_p(sp + off), per_buf))
        else:
            client._s.check(lib.mi355_probe_memory_copy(ctx, st, C.c_void_p(sp + off), C.c_void_p(dp + off), per_buf))

    t = _Timer(client)
    try:
        for i in range(3):
            one(i)
        client.sync()
        passes, secs = 8, 0.0
        while True:
            secs = t.run(one, passes)
            if secs >= min_seconds or passes >= 1 << 16:
                break
            passes *= 4
        return per_buf * access.buffers() * passes / secs
    finally:
        t.close()


def measure_memory_curve(client, access: MemoryAccess, *, cap: Optional[int] = None) -> MemoryCurve:
    """std/throughput/base.rs:46-64.  The sweep starts at the smallest working set the probes can move (one 32 KiB
    tile per buffer); `ceiling_at` clamps below it, as the reference's curve does below its first point."""
    max_alloc = int(client.properties().max_page_size)
    cap = cap if cap is not None else min(DEFAULT_BUFFER_BYTES, max_alloc) * access.buffers()
    floor = PROBE_TILE_BYTES * access.buffers()
    points = [MemoryPoint(b, measure_working_set(client, access, b)) for b in working_set_sweep(cap) if b >= floor]
    client.memory_cleanup()
    return MemoryCurve(access, points)


@dataclass(frozen=True)
class Work:
    """tune `Work`: what one launch has to do."""
    compute_ops: int
    bytes: int


@dataclass(frozen=True)
class Thresholds:
    """bounds_generator.rs:86-103: fraction of each peak a good kernel is expected to reach."""
    compute: float
    memory: float

    @staticmethod
    def uniform(fraction: float) -> "Thresholds":
        return Thresholds(fraction, fraction)


@dataclass(frozen=True)
class Bounds:
    compute_ops_per_s: float
    memory_bytes_per_s: float
    launch_overhead_s: float
    work: Work
    thresholds: Thresholds

    def time_limit(self) -> Optional[float]:
        """bounds_generator.rs:137-160: the slower of the two resource bounds at its threshold, plus one launch."""
        limits = []
        for amount, peak, thr in ((self.work.compute_ops, self.compute_ops_per_s, self.thresholds.compute),
                                  (self.work.bytes, self.memory_bytes_per_s, self.thresholds.memory)):
            if not (thr > 0.0 and thr < float("inf")) or not (peak > 0.0):
                continue
            limits.append(amount / peak / thr)
        if not limits:
            return None
        return max(limits) + self.launch_overhead_s


def measure_compute(client, dtype: int = N.DTYPE_BF16, iters: int = 20000) -> float:
    """Matrix-pipe issue rate (compute_cmma runner) in FLOP/s; dtype F32 / BF16 / F16 / F8E4M3."""
    sink = client.empty(256)
    n_ops = C.c_uint64()
    call = lambda _i: client._s.check(client.lib.mi355_probe_mfma(client.ctx, client.stream, dtype, iters,
                                                                  C.c_void_p(sink.device_ptr()), C.byref(n_ops)))
    t = _Timer(client)
    try:
        call(0)
        client.sync()
        secs = t.run(call, 4)
        return n_ops.value * 4 / secs
    finally:
        t.close()


def measure_launch_overhead(client, launches: int = 1000) -> float:
    """Seconds per empty launch (launch_overhead runner)."""
    sink = client.empty(256)
    call = lambda _i: client._s.check(client.lib.mi355_probe_launch_overhead(client.ctx, client.stream, launches,
                                                                            C.c_void_p(sink.device_ptr())))
    t = _Timer(client)
    try:
        call(0)
        client.sync()
        return t.run(call, 1) / launches
    finally:
        t.close()


def roofline_bounds(client, work: Work, thresholds: Thresholds, *, dtype: int = N.DTYPE_BF16,
                    curve: Optional[MemoryCurve] = None) -> Bounds:
    """std/throughput/base.rs:147-170; with a curve the memory ceiling is the one for this launch's working set."""
    mem = curve.ceiling_at(work.bytes) if curve is not None else measure_working_set(client, MemoryAccess.Copy,
                                                                                     2 * DEFAULT_BUFFER_BYTES)
    return Bounds(measure_compute(client, dtype), float(mem or 0.0), measure_launch_overhead(client), work, thresholds)


# ---- keyed peaks through the reference's sampling protocol ------------------------------------------------------------
_MFMA_PROBE_TILE = {N.DTYPE_BF16: (32, 32, 16), N.DTYPE_F16: (32, 32, 16), N.DTYPE_F32: (32, 32, 2), N.DTYPE_F8E4M3: (32, 32, 64)}


def _kernel_config(client, key: ThroughputKey, keep: list) -> Optional[KernelConfig]:
    """The `build_kernel` of each runner (std/throughput/runners/*.rs): buffers + a closure that enqueues `iterations`
    passes and returns the device seconds they took.  `ops_count` follows the reference's units: operations for the
    compute keys, F32 ELEMENTS for the memory keys (`bytes_per_s` multiplies by the key's element size), launches for
    the launch key.  None when this device has no probe for the key."""
    lib, ctx, st = client.lib, client.ctx, client.stream
    mode = key.mode
    t = _Timer(client)
    keep.append(t)
    sink = client.empty(256)
    sk = C.c_void_p(sink.device_ptr())
    keep.append(sink)
    if mode.kind in ("ComputeDirect", "ComputeCmma"):
        iters, n_ops = 4096, C.c_uint64()
        if mode.kind == "ComputeDirect":
            if mode.dtype != N.DTYPE_F32:
                return None
            call = lambda _i: client._s.check(lib.mi355_probe_compute_direct(ctx, st, iters, sk, C.byref(n_ops)))
        else:
            d = mode.config.cmma_dims
            if _MFMA_PROBE_TILE.get(mode.dtype) != (d.m, d.n, d.k) or mode.config.accumulator_type != N.DTYPE_F32:
                return None
            call = lambda _i: client._s.check(lib.mi355_probe_mfma(ctx, st, mode.dtype, iters, sk, C.byref(n_ops)))
        call(0)                                               # fills n_ops
        return KernelConfig(lambda iterations: t.run(call, iterations), int(n_ops.value))
    if mode.kind == "Launch":
        call = lambda _i: client._s.check(lib.mi355_probe_launch_overhead(ctx, st, 1, sk))
        return KernelConfig(lambda iterations: t.run(call, iterations), 1)
    access, working_set = mode.memory_probe()
    per_buf = max(working_set // access.buffers(), PROBE_TILE_BYTES) // PROBE_TILE_BYTES * PROBE_TILE_BYTES
    pool_bytes = max(DEFAULT_BUFFER_BYTES, per_buf)
    windows = max(pool_bytes // per_buf, 1)
    src = client.empty(pool_bytes)
    dst = client.empty(pool_bytes) if access is MemoryAccess.Copy else None
    keep.extend([src, dst])
    client._s.check(lib.mi355_memset(ctx, st, C.c_void_p(src.device_ptr()), 0, pool_bytes))
    sp, dp = src.device_ptr(), (dst.device_ptr() if dst is not None else 0)
    cursor = [0]

    def one(_i):
        off = (cursor[0] % windows) * per_buf                 # the window keeps moving across samples: cold every pass
        cursor[0] += 1
        if access is MemoryAccess.Read:
            client._s.check(lib.mi355_probe_memory_read(ctx, st, C.c_void_p(sp + off), per_buf, 1, sk))
        elif access is MemoryAccess.Write:
            client._s.check(lib.mi355_probe_memory_write(ctx, st, C.c_void_p(sp + off), per_buf))
        else:
            client._s.check(lib.mi355_probe_memory_copy(ctx, st, C.c_void_p(sp + off), C.c_void_p(dp + off), per_buf))
    return KernelConfig(lambda iterations: t.run(one, iterations), per_buf * access.buffers() // 4)


def measure_peak_throughput(client, key: ThroughputKey, *, cache_enabled: Optional[bool] = None) -> ThroughputValue:
    """std/throughput/base.rs:77-141: the peak this device attains for `key`, by the reference's protocol (plateau
    warm-up, best of 20-200 samples), cached per device name; `ThroughputValue.ZERO` where no probe exists (the
    reference returns it for a cmma key on a device without matrix instructions)."""
    keep: list = []
    try:
        cfg = _kernel_config(client, key, keep)
        if cfg is None:
            return ThroughputValue.ZERO
        return client.measure_throughput(key, cfg, cache_enabled=cache_enabled)
    finally:
        for k in keep:
            if isinstance(k, _Timer):
                k.close()
        keep.clear()
        client.memory_cleanup()


def device_throughput(client, keys: Iterable[ThroughputKey]) -> List[ThroughputValue]:
    """std/throughput/base.rs:30-44."""
    return [measure_peak_throughput(client, k) for k in keys]
