"""The reference's benchmark protocol: `Benchmark`, `BenchmarkDurations`, `BenchmarkComputations`,
`run_benchmark`.

Mirrors crates/cubecl-common/src/benchmark.rs:
  * `BenchmarkDurations`      :15-54    raw samples + the timing method; mean / variance / min / max / median
  * `Display`                 :56-86    the "Result" box every reference bench prints
  * `BenchmarkComputations`   :88-161   the five statistics and `score()` (lower is better; what autotune ranks by)
  * `Benchmark`               :163-300  prepare / execute / sync / profile; `run` = 5 warm-up executions, then
                                        `num_samples()` (15, or BENCH_NUM_SAMPLES) profiled executions
  * `run_benchmark`           :327-356  wall-clock run + name / options / shapes / git hash / timestamp

Durations are integer nanoseconds, as `core::time::Duration` holds them: the mean truncates (`Duration / u32`), each
squared deviation is rounded to the nearest nanosecond (`Duration::from_secs_f64`), so the statistics are the
reference's to the last digit -- tests/test_host_logic.py holds the reference's own known answers (:359-428).

`DeviceBenchmark` is the one addition: what every reference bench writes by hand (`fn sync` = `client.sync()`,
`fn profile` = `client.profile(|| self.execute(args))`), here over `ComputeClient.profile` (HIP events on the
client's stream, `mi355_profile_start / _stop`).
"""
from __future__ import annotations

import enum
import math
import os
import subprocess
import time
from dataclasses import dataclass, field
from typing import Any, List, Optional, Sequence

NANOS_PER_SEC = 1_000_000_000


class TimingMethod(enum.Enum):
    """profile.rs `TimingMethod`: wall clock around sync points, or the device's own timestamps."""
    System = "system"
    Device = "device"

    def __str__(self) -> str:
        return self.value


def duration_from_secs_f64(secs: float) -> int:
    """`Duration::from_secs_f64`: nearest nanosecond, ties to even; refuses what a Duration cannot hold."""
    if not math.isfinite(secs) or secs < 0.0:
        raise ValueError(f"cannot convert {secs!r} seconds to a Duration")
    whole = math.floor(secs)
    frac_ns = (secs - whole) * NANOS_PER_SEC        # exact: one subtraction of nearby doubles, then < 2^30
    lo = math.floor(frac_ns)
    rem = frac_ns - lo
    if rem > 0.5 or (rem == 0.5 and lo % 2 == 1):
        lo += 1
    return int(whole) * NANOS_PER_SEC + int(lo)


def format_duration(nanos: int, precision: int = 3) -> str:
    """`{:.N?}` of a Duration: the largest unit that leaves a non-zero integer part (s, ms, µs, ns), the fraction
    rounded to N digits (ties to even) with carry into the integer part."""
    if nanos >= NANOS_PER_SEC:
        unit, div = "s", NANOS_PER_SEC
    elif nanos >= 1_000_000:
        unit, div = "ms", 1_000_000
    elif nanos >= 1_000:
        unit, div = "µs", 1_000
    else:
        unit, div = "ns", 1
    integer, rest = divmod(nanos, div)
    scale = 10 ** precision
    frac, rem = divmod(rest * scale, div)
    last_is_odd = (frac if precision else integer) % 2 == 1
    if rem * 2 > div or (rem * 2 == div and last_is_odd):          # ties to even, as core::time prints them
        frac += 1
        if frac == scale:
            frac, integer = 0, integer + 1
    return f"{integer}.{frac:0{precision}d}{unit}" if precision else f"{integer}{unit}"


@dataclass
class BenchmarkDurations:
    """benchmark.rs:15-54.  `durations` in nanoseconds."""
    timing_method: TimingMethod
    durations: List[int]

    @staticmethod
    def from_durations(timing_method: TimingMethod, durations: Sequence[int]) -> "BenchmarkDurations":
        return BenchmarkDurations(timing_method, list(durations))

    def min_max_median_durations(self):
        s = sorted(self.durations)
        return s[0], s[-1], s[len(s) // 2]          # upper median for an even count, as the reference takes it

    def mean_duration(self) -> int:
        return sum(self.durations) // len(self.durations)

    def variance_duration(self, mean: int) -> int:
        """Population variance, carried in a Duration (so: seconds squared, stored as if they were seconds)."""
        total = 0
        for d in self.durations:
            tmp = d / NANOS_PER_SEC - mean / NANOS_PER_SEC
            total += duration_from_secs_f64(tmp * tmp)
        return total // len(self.durations)

    def __str__(self) -> str:
        c = BenchmarkComputations.new(self)
        return ("\n―――――――― Result ―――――――――\n"
                f"  Timing      {self.timing_method}\n"
                f"  Samples     {len(self.durations)}\n"
                f"  Mean        {format_duration(c.mean)}\n"
                f"  Variance    {format_duration(c.variance)}\n"
                f"  Median      {format_duration(c.median)}\n"
                f"  Min         {format_duration(c.min)}\n"
                f"  Max         {format_duration(c.max)}\n"
                "―――――――――――――――――――――――――")


@dataclass
class BenchmarkComputations:
    """benchmark.rs:88-161."""
    mean: int = 0
    median: int = 0
    variance: int = 0
    min: int = 0
    max: int = 0

    @staticmethod
    def new(durations: BenchmarkDurations) -> "BenchmarkComputations":
        mean = durations.mean_duration()
        lo, hi, median = durations.min_max_median_durations()
        return BenchmarkComputations(mean=mean, median=median, min=lo, max=hi,
                                     variance=durations.variance_duration(mean))

    def score(self) -> int:
        """:117-160 -- 0.8 of the fastest run + 0.2 of the median, inflated by the coefficient of variation."""
        alpha = 0.8
        base = self.min * alpha + self.median * (1.0 - alpha)
        std_dev = math.sqrt(float(self.variance))
        return int(base * (1.0 + std_dev / (1.0 + float(self.mean))))


class Benchmark:
    """benchmark.rs:163-300.  Subclasses give `prepare`, `execute`, `name`, `sync`; the rest has the reference's defaults."""

    WARMUP_EXECUTIONS = 5           # :268-275: "the first one probably triggers the JIT-compilation", then 4 warm-ups
    DEFAULT_NUM_SAMPLES = 15        # :183

    def prepare(self) -> Any:
        raise NotImplementedError

    def execute(self, input: Any) -> Any:
        raise NotImplementedError

    def name(self) -> str:
        raise NotImplementedError

    def sync(self) -> None:
        raise NotImplementedError

    def num_samples(self) -> int:
        raw = os.environ.get("BENCH_NUM_SAMPLES", "")
        return int(raw) if raw.isascii() and raw.isdigit() else self.DEFAULT_NUM_SAMPLES   # parse::<usize> or the default

    def options(self) -> Optional[str]:
        return None

    def shapes(self) -> List[List[int]]:
        return []

    def work(self):
        """tune `Work` of one execution (cubecl_amd.throughput.Work), when the bench can state it."""
        return None

    def profile(self, args: Any) -> int:
        """Nanoseconds of one execution by the device's clock; the default is the wall-clock `profile_full`."""
        return self.profile_full(args)

    def profile_full(self, args: Any) -> int:
        self.sync()
        start = time.perf_counter_ns()
        out = self.execute(args)
        self.sync()
        del out
        return time.perf_counter_ns() - start

    def run(self, timing_method: TimingMethod) -> BenchmarkDurations:
        one = self.profile_full if timing_method is TimingMethod.System else self.profile
        args = self.prepare()
        for _ in range(self.WARMUP_EXECUTIONS):
            try:
                one(args)                   # a failing warm-up is not fatal in the reference either (:272-274)
            except Exception:               # noqa: BLE001
                pass
        return BenchmarkDurations(timing_method, [one(args) for _ in range(self.num_samples())])


class DeviceBenchmark(Benchmark):
    """A `Benchmark` over a `ComputeClient`: `sync` and the device-clock `profile` as the reference's benches write them."""

    def __init__(self, client):
        self.client = client

    def sync(self) -> None:
        self.client.sync()

    def profile(self, args: Any) -> int:
        _, nanos = self.client.profile(lambda: self.execute(args), self.name())
        return nanos


@dataclass
class BenchmarkResult:
    """benchmark.rs:302-325."""
    raw: BenchmarkDurations
    computed: BenchmarkComputations
    git_hash: str
    name: str
    options: Optional[str]
    shapes: List[List[int]]
    timestamp: int
    extra: dict = field(default_factory=dict)

    def __str__(self) -> str:
        return (f"\n        Timestamp: {self.timestamp}\n        Git Hash: {self.git_hash}\n"
                f"        Benchmarking - {self.name}{self.raw}\n        ")


def run_benchmark(benchmark: Benchmark, timing_method: TimingMethod = TimingMethod.System) -> BenchmarkResult:
    """:327-356.  The reference always takes the wall clock here; a device-clock run is one argument away."""
    timestamp = time.time_ns() // 1_000_000
    try:
        git_hash = subprocess.run(["git", "rev-parse", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip()
    except Exception:                       # noqa: BLE001  (no git on the box: the hash is a label, not a dependency)
        git_hash = ""
    durations = benchmark.run(timing_method)
    return BenchmarkResult(raw=durations, computed=BenchmarkComputations.new(durations), git_hash=git_hash,
                           name=benchmark.name(), options=benchmark.options(), shapes=benchmark.shapes(),
                           timestamp=timestamp)
