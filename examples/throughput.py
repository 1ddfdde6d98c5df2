"""The reference's `examples/throughput` (examples/throughput/src/lib.rs, examples/*.rs) on the MI355X path.

    python examples/throughput.py [all | compute_direct | compute_cmma | memory | memory_read | memory_write |
                                   launch_overhead | memory_curve | roofline]

Every line is one `ThroughputKey` measured by `measure_peak_throughput` (plateau warm-up, best of 20-200 samples, cached per
device) and printed with `ThroughputValue.format`, as the reference's `run` prints it (lib.rs:125-140).  `roofline` is
ours: the time limit the measured ceilings imply for one 8192^3 bf16 GEMM (`roofline_bounds`, std/throughput/base.rs:147-170).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cubecl_amd import ElemType, Mi355Runtime  # noqa: E402
from cubecl_amd import _native as N  # noqa: E402
from cubecl_amd import roofline as R  # noqa: E402
from cubecl_amd import throughput as T  # noqa: E402

K, M = R.ThroughputKey, R.ThroughputMode


def cmma_keys(client):
    """lib.rs `compute_cmma_key` asks for one f16 tile; this device has a matrix instruction per input type."""
    out = []
    for dt in (N.DTYPE_BF16, N.DTYPE_F16, N.DTYPE_F32, N.DTYPE_F8E4M3):
        tile = R.select_cmma_tile(client.features()["cmma"], dt, dt, N.DTYPE_F32, (8192, 8192, 8192))
        if tile is not None:
            out.append(R.compute_throughput_key(tile, dt, N.DTYPE_F32))
    return out


def bytes_label(nbytes: int) -> str:
    value, unit = float(nbytes), 0
    while value >= 1024.0 and unit < 3:
        value /= 1024.0
        unit += 1
    return f"{value:.0f} {('B', 'KiB', 'MiB', 'GiB')[unit]}"


def describe(key) -> str:
    m = key.mode
    if m.kind == "ComputeCmma":
        d = m.config.cmma_dims
        return f"{ElemType(m.dtype).name.lower()}→{ElemType(m.config.accumulator_type).name.lower()} {d.m}×{d.n}×{d.k}"
    if m.kind == "ComputeDirect":
        return ElemType(key.dtype()).name.lower()
    if m.kind == "MemoryWorkingSet":
        return bytes_label(m.bytes)
    return ""


def run(client, keys) -> None:
    print(f"Peak throughput — {Mi355Runtime.name()}")
    for key in keys:
        value = T.measure_peak_throughput(client, key).format(key)
        print(f"  {key.mode.kind:<15}{describe(key):<24}{value:>18}")


def memory_curve(client) -> None:
    print(f"Memory curve — {Mi355Runtime.name()}")
    for access in (R.MemoryAccess.Read, R.MemoryAccess.Write, R.MemoryAccess.Copy):
        curve = T.measure_memory_curve(client, access)
        print(f"\n  {access.value}")
        for pt in curve.points():
            print(f"    {bytes_label(pt.bytes):>10}{curve.ceiling_at(pt.bytes) / 1e9:>9.1f} GB/s")


def roofline(client) -> None:
    S = 8192
    b = T.roofline_bounds(client, T.Work(2 * S ** 3, 3 * S * S * 2), T.Thresholds.uniform(0.5))
    print(f"roofline time limit for one {S}^3 bf16 GEMM at 50 % of both ceilings: {b.time_limit() * 1e3:.3f} ms")


def main() -> None:
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    client = Mi355Runtime.client()
    p = client.properties()
    print(f"device: {p.name.decode()} {p.gcn_arch_name.decode()}, {p.num_streaming_multiprocessors} CUs, {p.total_memory / 2**30:.0f} GiB")
    table = {
        "compute_direct": lambda: [K(M.ComputeDirect(N.DTYPE_F32))],
        "compute_cmma": lambda: cmma_keys(client),
        "memory": lambda: [K(M.Memory)],
        "memory_read": lambda: [K(M.MemoryRead)],
        "memory_write": lambda: [K(M.MemoryWrite)],
        "launch_overhead": lambda: [K(M.Launch)],
    }
    if what == "all":
        run(client, [k for name in ("compute_direct", "compute_cmma", "memory", "memory_read", "memory_write", "launch_overhead")
                     for k in table[name]()])
    elif what in table:
        run(client, table[what]())
    elif what == "memory_curve":
        memory_curve(client)
    elif what == "roofline":
        roofline(client)
    else:
        raise SystemExit(__doc__)


if __name__ == "__main__":
    main()
