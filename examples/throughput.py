"""The reference's `examples/throughput` (examples/throughput/src/lib.rs) on the MI355X path: the measured ceilings of
the device -- copy / read / write bandwidth, the read working-set curve, matrix-pipe issue rates per type, launch
overhead -- and the roofline time limit they imply for one 8192^3 bf16 GEMM.

    python examples/throughput.py [--curve]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cubecl_amd import Mi355Runtime  # noqa: E402
from cubecl_amd import _native as N  # noqa: E402
from cubecl_amd import throughput as T  # noqa: E402


def main() -> None:
    client = Mi355Runtime.client()
    p = client.properties()
    print(f"device: {p.name.decode()} {p.gcn_arch_name.decode()}, {p.num_streaming_multiprocessors} CUs, {p.total_memory / 2**30:.0f} GiB")
    for access, ws in ((T.MemoryAccess.Copy, 1 << 30), (T.MemoryAccess.Read, 512 << 20), (T.MemoryAccess.Write, 512 << 20)):
        print(f"memory {access.value:5s}: {T.measure_working_set(client, access, ws) / 1e12:6.2f} TB/s   (working set {ws >> 20} MiB)")
    for name, dt in (("f32", N.DTYPE_F32), ("bf16", N.DTYPE_BF16), ("f16", N.DTYPE_F16), ("fp8 e4m3", N.DTYPE_F8E4M3)):
        print(f"matrix pipe {name:9s}: {T.measure_compute(client, dt) / 1e12:8.1f} TFLOP/s")
    print(f"launch overhead: {T.measure_launch_overhead(client) * 1e6:.2f} us")
    curve = None
    if "--curve" in sys.argv:
        curve = T.measure_memory_curve(client, T.MemoryAccess.Read)
        for pt in curve.points():
            print(f"  read, working set {pt.bytes >> 10:9d} KiB: {pt.bytes_per_s / 1e9:8.1f} GB/s")
    S = 8192
    b = T.roofline_bounds(client, T.Work(2 * S ** 3, 3 * S * S * 2), T.Thresholds.uniform(0.5), curve=curve)
    print(f"roofline time limit for one {S}^3 bf16 GEMM at 50 % of both ceilings: {b.time_limit() * 1e3:.3f} ms")


if __name__ == "__main__":
    main()
