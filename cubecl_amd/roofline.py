"""Roofline scoring, the peak-throughput sampling protocol and the throughput keys.

Mirrors crates/cubecl-runtime/src/throughput/:
  * roofline.rs:14-107     `ResourceBound.time_at_peak`, `binding_resource`, `score_resources`, `binding_achieved`
  * benchmarker.rs:13-143  `KernelConfig`, `ThroughputBenchmarker.measure` = warm up until the per-iteration time
                           plateaus (choosing the iteration count so that one sample lasts >= 20 ms), then the best of
                           20...200 samples with patience 12
  * cache.rs:11-73         `ThroughputCache.get_for_device` -- one store per device name, shared by its clients
  * base.rs:10-214         `MemoryAccess`, `ThroughputMode`, `ThroughputKey`, `ThroughputValue` (+ `format`),
                           `compute_throughput_key`
  * cmma.rs:4-60           `CmmaDims`, `ComputeCmmaConfig`, `select_cmma_tile`
and crates/cubecl-runtime/src/tune/bounds_generator.rs:14-151: `AutotuneBound`, `calculate_bounds`, the `time_limit`
reductions (`bounds_time_limit`, `TuneBounds`).

Pure host logic: no device, no library call.  Element types are this package's `ElemType` integers, so only the keys
without a type (`{"mode":"Memory"}` ..., the forms the reference pins in base.rs:218-240) serialise identically.  Durations are seconds (float) except where the reference's arithmetic
goes through a `Duration` and its nanosecond rounding is visible (`ThroughputValue.duration_per_op`).
`bench.py` prices its `roofline` objects with `score_resources`; `cubecl_amd.throughput.measure_peak_throughput`
drives the library's probes through `ThroughputBenchmarker`.
"""
from __future__ import annotations

import enum
import json
import math
import threading
from dataclasses import dataclass
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple

from .benchmark import NANOS_PER_SEC, duration_from_secs_f64, format_duration

DEFAULT_BUFFER_BYTES = 512 * 1024 * 1024      # base.rs:9


# ---- roofline.rs ------------------------------------------------------------------------------------------------------
def _is_normal(x: float) -> bool:
    """f64::is_normal: not zero, subnormal, infinite or NaN (the sign does not matter)."""
    return math.isfinite(x) and abs(x) >= 2.2250738585072014e-308


@dataclass(frozen=True)
class ResourceBound:
    """How much of a resource a run must move (bytes or operations) against the peak rate of that resource."""
    amount: int
    peak_per_s: float

    def time_at_peak(self) -> Optional[float]:
        """Seconds `amount` takes at `peak_per_s`; None for a peak that is zero, NaN or infinite.  A negative
        peak has no Duration either (the reference panics in `Duration::from_secs_f64`; here it is None)."""
        if not _is_normal(self.peak_per_s) or self.peak_per_s < 0.0:
            return None
        return self.amount / self.peak_per_s


def binding_resource(bounds: Sequence[ResourceBound]) -> Optional[ResourceBound]:
    """The resource needing the most time even at its own peak (the last of equals, as `max_by_key` returns it)."""
    best = None
    for b in bounds:
        t = b.time_at_peak()
        if t is not None and (best is None or t >= best[0]):
            best = (t, b)
    return best[1] if best else None


@dataclass(frozen=True)
class AchievedThroughput:
    achieved_per_s: float
    fraction_of_peak: float        # not clamped: beating the modelled peak is a finding about the model


def score_resources(duration_s: float, bounds: Sequence[ResourceBound]) -> List[AchievedThroughput]:
    """One measured duration against every bound, in `bounds` order; a zero duration gives NaN rates."""
    out = []
    for b in bounds:
        achieved = float("nan") if duration_s == 0.0 else b.amount / duration_s
        try:
            frac = achieved / b.peak_per_s
        except ZeroDivisionError:                      # IEEE: x / 0 = inf (NaN for NaN / 0)
            frac = float("nan") if achieved != achieved or achieved == 0.0 else math.copysign(float("inf"), achieved)
        out.append(AchievedThroughput(achieved, frac))
    return out


def binding_achieved(scores: Sequence[AchievedThroughput]) -> Optional[AchievedThroughput]:
    """The resource that governed the run: the largest finite fraction of peak (the last of equals)."""
    best = None
    for s in scores:
        if math.isfinite(s.fraction_of_peak) and (best is None or s.fraction_of_peak >= best.fraction_of_peak):
            best = s
    return best


# ---- tune/bounds_generator.rs: the roofline as a time limit ------------------------------------------------------------
@dataclass(frozen=True)
class AutotuneBound:
    """bounds_generator.rs:63-137: a resource bound and the fraction of its peak a good kernel is expected to reach."""
    resource: ResourceBound
    threshold: float

    def time_limit(self) -> Optional[float]:
        """time at peak / threshold; None when the threshold (or the peak) is zero, NaN or infinite."""
        if not _is_normal(self.threshold):
            return None
        t = self.resource.time_at_peak()
        return None if t is None else t / self.threshold


def bounds_time_limit(bounds: Sequence[AutotuneBound]) -> Optional[float]:
    """:139-143 -- resources overlap at best, so the achievable floor is the SLOWER bound: max, not min."""
    limits = [t for t in (b.time_limit() for b in bounds) if t is not None]
    return max(limits) if limits else None


@dataclass(frozen=True)
class TuneBounds:
    """bounds_generator.rs:14-24 `Bounds`: the per-resource bounds of one launch plus the launch overhead."""
    bounds: Tuple[AutotuneBound, ...]
    launch_overhead: float = 0.0                 # seconds

    def time_limit(self) -> Optional[float]:
        """:145-151 -- no usable bound means no limit at all: the launch overhead is not a limit on its own."""
        limit = bounds_time_limit(self.bounds)
        return None if limit is None else limit + self.launch_overhead


def calculate_bounds(work, thresholds, compute_throughput: "ThroughputValue", memory_throughput: "ThroughputValue",
                     memory_key: "ThroughputKey") -> List[AutotuneBound]:
    """:104-127.  `work` has `compute_ops` and `bytes` (cubecl_common::work::Work), `thresholds` has `compute` and `memory`."""
    return [AutotuneBound(ResourceBound(work.compute_ops, compute_throughput.ops_per_s()), thresholds.compute),
            AutotuneBound(ResourceBound(work.bytes, memory_throughput.bytes_per_s(memory_key)), thresholds.memory)]


# ---- base.rs / cmma.rs: keys and values -------------------------------------------------------------------------------
class MemoryAccess(enum.Enum):
    Copy = "Copy"
    Read = "Read"
    Write = "Write"

    def buffers(self) -> int:
        return 2 if self is MemoryAccess.Copy else 1

    def default_working_set(self) -> int:
        return DEFAULT_BUFFER_BYTES * self.buffers()


@dataclass(frozen=True)
class CmmaDims:
    m: int
    n: int
    k: int

    def num_elems(self) -> int:
        return self.m * self.n * self.k


@dataclass(frozen=True)
class ComputeCmmaConfig:
    accumulator_type: int          # ElemType
    cmma_dims: CmmaDims


@dataclass(frozen=True)
class ThroughputMode:
    """The reference's enum as a tagged record: `kind` is the variant, the other fields are that variant's payload."""
    kind: str                                      # ComputeDirect | ComputeCmma | Memory | MemoryRead | MemoryWrite | MemoryWorkingSet | Launch
    dtype: Optional[int] = None                    # ComputeDirect, ComputeCmma
    config: Optional[ComputeCmmaConfig] = None     # ComputeCmma
    access: Optional[MemoryAccess] = None          # MemoryWorkingSet
    bytes: Optional[int] = None                    # MemoryWorkingSet

    @staticmethod
    def ComputeDirect(dtype: int) -> "ThroughputMode":
        return ThroughputMode("ComputeDirect", dtype=int(dtype))

    @staticmethod
    def ComputeCmma(dtype: int, config: ComputeCmmaConfig) -> "ThroughputMode":
        return ThroughputMode("ComputeCmma", dtype=int(dtype), config=config)

    @staticmethod
    def MemoryWorkingSet(access: MemoryAccess, nbytes: int) -> "ThroughputMode":
        return ThroughputMode("MemoryWorkingSet", access=access, bytes=int(nbytes))

    def memory_probe(self) -> Optional[Tuple[MemoryAccess, int]]:
        """base.rs:62-80: the one place the memory modes map onto (access, working set)."""
        fixed = {"Memory": MemoryAccess.Copy, "MemoryRead": MemoryAccess.Read, "MemoryWrite": MemoryAccess.Write}
        if self.kind in fixed:
            return fixed[self.kind], fixed[self.kind].default_working_set()
        if self.kind == "MemoryWorkingSet":
            return self.access, self.bytes
        return None


ThroughputMode.Memory = ThroughputMode("Memory")
ThroughputMode.MemoryRead = ThroughputMode("MemoryRead")
ThroughputMode.MemoryWrite = ThroughputMode("MemoryWrite")
ThroughputMode.Launch = ThroughputMode("Launch")

_F32 = 0            # _native.DTYPE_F32; spelled out so that this module needs no library (checked in tests/test_host_logic.py)
_DTYPE_BYTES = {0: 4, 1: 2, 2: 2, 3: 8, 4: 4, 5: 4, 6: 8, 7: 8, 8: 1, 9: 1, 10: 1, 11: 1, 12: 1, 13: 1, 14: 2, 15: 2, 16: 1, 17: 4}


@dataclass(frozen=True)
class ThroughputKey:
    mode: ThroughputMode

    def dtype(self) -> int:
        """base.rs:93-106: the element type of a compute key; F32 for the memory and launch keys."""
        return self.mode.dtype if self.mode.kind in ("ComputeDirect", "ComputeCmma") else _F32

    def to_json(self) -> str:
        """The serialised form the reference's persistent cache keys on (serde's externally tagged enum)."""
        m = self.mode
        if m.kind == "ComputeDirect":
            mode = {"ComputeDirect": {"dtype": m.dtype}}
        elif m.kind == "ComputeCmma":
            d = m.config.cmma_dims
            mode = {"ComputeCmma": {"dtype": m.dtype, "config": {"accumulator_type": m.config.accumulator_type,
                                                                 "cmma_dims": {"m": d.m, "n": d.n, "k": d.k}}}}
        elif m.kind == "MemoryWorkingSet":
            mode = {"MemoryWorkingSet": {"access": m.access.value, "bytes": m.bytes}}
        else:
            mode = m.kind
        return json.dumps({"mode": mode}, separators=(",", ":"))

    @staticmethod
    def from_json(text: str) -> "ThroughputKey":
        obj = json.loads(text)
        if set(obj) != {"mode"}:
            raise ValueError("unknown field in a ThroughputKey")         # serde(deny_unknown_fields), base.rs:85
        mode = obj["mode"]
        if isinstance(mode, str):
            if mode not in ("Memory", "MemoryRead", "MemoryWrite", "Launch"):
                raise ValueError(f"unknown ThroughputMode {mode!r}")
            return ThroughputKey(ThroughputMode(mode))
        (kind, body), = mode.items()
        if kind == "ComputeDirect":
            return ThroughputKey(ThroughputMode.ComputeDirect(body["dtype"]))
        if kind == "ComputeCmma":
            c = body["config"]
            return ThroughputKey(ThroughputMode.ComputeCmma(body["dtype"], ComputeCmmaConfig(
                c["accumulator_type"], CmmaDims(**c["cmma_dims"]))))
        if kind == "MemoryWorkingSet":
            return ThroughputKey(ThroughputMode.MemoryWorkingSet(MemoryAccess(body["access"]), body["bytes"]))
        raise ValueError(f"unknown ThroughputMode {kind!r}")


@dataclass(frozen=True)
class ThroughputValue:
    """base.rs:110-186.  `duration` in seconds for `ops_count` operations (elements, for the memory keys)."""
    ops_count: int
    duration: float

    def ops_per_s(self) -> float:
        return float("nan") if self.duration == 0.0 else self.ops_count / self.duration

    def bytes_per_s(self, key: ThroughputKey) -> float:
        return float("nan") if self.duration == 0.0 else self.ops_count * _DTYPE_BYTES[key.dtype()] / self.duration

    def duration_per_op(self) -> float:
        """Seconds per operation, to the nanosecond as the reference's Duration carries it."""
        if self.ops_count == 0:
            return 0.0
        return duration_from_secs_f64(self.duration / self.ops_count) / NANOS_PER_SEC

    def format(self, key: ThroughputKey) -> str:
        kind = key.mode.kind
        if kind == "Launch":
            nanos = duration_from_secs_f64(self.duration / self.ops_count) if self.ops_count else 0
            return "N/A" if nanos == 0 else f"{_debug_duration(nanos)}/launch"
        val, unit = (self.ops_per_s(), "OPS") if kind in ("ComputeDirect", "ComputeCmma") else (self.bytes_per_s(key), "bytes")
        if val != val:
            return "N/A"
        suffixes = ["", "K", "M", "G", "T", "P", "E", "Z", "Y", "R", "Q"]
        idx = 0
        for _ in range(len(suffixes) - 1):
            if val < 1000.0:
                break
            val /= 1000.0
            idx += 1
        return f"{val:.4f} {suffixes[idx]}{unit}/s"


ThroughputValue.ZERO = ThroughputValue(0, 0.0)


def _debug_duration(nanos: int) -> str:
    """`{:?}` of a Duration: the shortest exact decimal in the largest unit (trailing zeros dropped)."""
    text = format_duration(nanos, 9)
    number = text.rstrip("sµmn")
    unit = text[len(number):]
    number = number.rstrip("0").rstrip(".")
    return number + unit


def compute_throughput_key(cmma_tile: Optional[Tuple[int, int, int]], input_elem_type: int, acc_elem_type: int) -> ThroughputKey:
    """base.rs:188-214: the cmma key for a tile, the direct (FMA) key on the accumulator type without one."""
    if cmma_tile is None:
        return ThroughputKey(ThroughputMode.ComputeDirect(acc_elem_type))
    m, n, k = cmma_tile
    return ThroughputKey(ThroughputMode.ComputeCmma(input_elem_type, ComputeCmmaConfig(int(acc_elem_type), CmmaDims(m, n, k))))


def select_cmma_tile(mma_configs: Iterable[Tuple[int, int, int, int, int, int]], lhs: int, rhs: int, acc: int,
                     problem: Tuple[int, int, int]) -> Optional[Tuple[int, int, int]]:
    """cmma.rs:31-60 over `ComputeClient.features()["cmma"] | ["mma"]` entries `(a, b, cd, m, n, k)`: the matrix
    instruction with exactly these types that fits inside the problem and has the largest volume.  The reference breaks
    ties by registration order (`max_by_key` keeps the last); `features()` is a set, so equal volumes are decided by the
    larger (m, n, k) instead -- f32 32x32x2 over 16x16x8."""
    pm, pn, pk = problem
    best = None
    for (a, b, cd, m, n, k) in mma_configs:
        if (a, b, cd) != (lhs, rhs, acc) or pm < m or pn < n or pk < k:
            continue
        if best is None or (m * n * k, m, n, k) > (best[0] * best[1] * best[2],) + best:
            best = (m, n, k)
    return best


# ---- cache.rs ---------------------------------------------------------------------------------------------------------
class ThroughputCache:
    """In-memory store per device name (the reference persists it when built with std_io; a process-lifetime store is
    what its no-io build has).  `insert` keeps an existing value, as the persistent store does on a concurrent write."""
    _global: Dict[str, "ThroughputCache"] = {}
    _lock = threading.Lock()

    def __init__(self, name: str = ""):
        self.name = name
        self._cache: Dict[ThroughputKey, ThroughputValue] = {}
        self.lock = threading.Lock()

    @classmethod
    def get_for_device(cls, name: str) -> "ThroughputCache":
        with cls._lock:
            return cls._global.setdefault(name, ThroughputCache(name))

    def insert(self, key: ThroughputKey, value: ThroughputValue) -> None:
        self._cache.setdefault(key, value)

    def get(self, key: ThroughputKey) -> Optional[ThroughputValue]:
        return self._cache.get(key)


def env_bool(name: str) -> Optional[bool]:
    """config/base.rs:186-192: true / 1 / on, false / 0 / off, anything else is no opinion."""
    import os
    return {"true": True, "1": True, "on": True, "false": False, "0": False, "off": False}.get(os.environ.get(name, ""))


def throughput_cache_enabled() -> bool:
    enabled = env_bool("CUBECL_THROUGHPUT_CACHE")
    return True if enabled is None else enabled


# ---- benchmarker.rs ---------------------------------------------------------------------------------------------------
@dataclass
class KernelConfig:
    sample: Callable[[int], float]          # runs the kernel `iterations` times, returns the seconds that took
    ops_count: int                          # operations of ONE iteration


class ThroughputBenchmarker:
    MAX_WARMUP = 50
    MAX_ITERATIONS = 1_000
    PLATEAU_TOL = 0.03
    WARMUP_PATIENCE = 3
    TARGET_DURATION_MS = 20.0
    MIN_SAMPLES = 20
    MAX_SAMPLES = 200
    REL_TOL = 0.01
    SAMPLE_PATIENCE = 12

    def __init__(self, cache: ThroughputCache, cache_enabled: Optional[bool] = None):
        """`cache_enabled=None` takes the configuration: on, unless CUBECL_THROUGHPUT_CACHE says off / 0 / false
        (config/base.rs:157-159, :186-192; examples/throughput/README.md "Caching")."""
        self.cache = cache
        self.cache_enabled = throughput_cache_enabled() if cache_enabled is None else cache_enabled

    def measure(self, key: ThroughputKey, kernel_config: KernelConfig) -> ThroughputValue:
        """benchmarker.rs:40-66: the peak attained -- the minimum time per iteration after the device has warmed up."""
        if self.cache_enabled:
            with self.cache.lock:
                hit = self.cache.get(key)
            if hit is not None:
                return hit
        iterations = self.warmup(kernel_config.sample)
        value = ThroughputValue(kernel_config.ops_count, self.sample_peak_duration(iterations, kernel_config.sample))
        if self.cache_enabled:
            with self.cache.lock:
                self.cache.insert(key, value)
        return value

    def warmup(self, sample: Callable[[int], float]) -> int:
        """:70-108.  Grows the iteration count until one sample lasts TARGET_DURATION_MS, then keeps sampling until the
        per-iteration time has stopped improving by more than PLATEAU_TOL for WARMUP_PATIENCE samples in a row."""
        best, stable, iterations = math.inf, 0, 1
        for _ in range(self.MAX_WARMUP):
            duration = sample(iterations) * 1000.0
            if duration < self.TARGET_DURATION_MS:
                if duration > 1e-6:
                    extra = math.ceil((self.TARGET_DURATION_MS - duration) / (duration / iterations))
                else:
                    extra = iterations
                iterations = min(iterations + max(extra, 1), self.MAX_ITERATIONS)
                best, stable = math.inf, 0
                continue
            per_iter = duration / iterations
            if per_iter < best * (1.0 - self.PLATEAU_TOL):
                best, stable = per_iter, 0
            else:
                best = min(best, per_iter)
                stable += 1
                if stable >= self.WARMUP_PATIENCE:
                    break
        return iterations

    def sample_peak_duration(self, iterations: int, sample_once: Callable[[int], float]) -> float:
        """:112-143.  Seconds per iteration of the best sample; stops once MIN_SAMPLES were taken and the best has not
        improved by more than REL_TOL for SAMPLE_PATIENCE samples."""
        assert iterations > 0, "iterations must be positive"
        best, stale = math.inf, 0
        for i in range(self.MAX_SAMPLES):
            s = sample_once(iterations)
            if s < best * (1.0 - self.REL_TOL):
                best, stale = s, 0
            else:
                best = min(best, s)
                stale += 1
            if i > self.MIN_SAMPLES and stale >= self.SAMPLE_PATIENCE:
                break
        return best / iterations
