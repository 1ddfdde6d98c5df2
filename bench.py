#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native GEMM / reduce hot path.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver starts one
process per GPU with torch.distributed.run.  Rank 0 prints ONE JSON line.

  metric  : BASELINE.json -- "GEMM TFLOP/s (8192^3 bf16) + reduce GB/s vs roofline"
  step    : one 8192 x 8192 x 8192 bf16 GEMM (f32 accumulate, bf16 C) per rank on synthetic
            operands already resident in HBM (config C3).  N ranks = a batch of N such GEMMs,
            batch-sharded one per GPU with no data-path collective => "scaling": "weak".
  value   : whole-job TFLOP/s = N * 2*8192^3 * K / (max over ranks of the timed region).
  roofline: the GEMM kernel against the dense bf16 MFMA peak (2.5 PFLOP/s), from HIP events
            recorded on the stream the kernel is launched on.
  cpu_baseline : the CPU restatement (oracle/, cubecl-cpu execution model) timed on this box's
            cores on a bounded sample of the same workload.  Reported, never the thing measured.
  extra   : the other BASELINE.json configs (f32 4096^3, 1 GiB sum/argmax (+ RCCL all-reduce when
            N > 1), batched 2048^3 shard, skinny GEMMs) and the measured HBM / MFMA ceilings.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on these hosts (RCCL across processes)
SEED = 0x5EEDC0BE
PEAK_BF16_TFLOPS = 2500.0   # dense, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
PEAK_F32_TFLOPS = 157.3     # MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_HBM_GBS = 8000.0       # HBM3E spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=8192, help="M=N=K of the headline GEMM (8192 = config C3)")
    ap.add_argument("--no-extras", action="store_true", help="headline + roofline + cpu_baseline only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-plateau-warmup", action="store_true", help="only the W warm-up steps (measures the DVFS ramp too)")
    ap.add_argument("--algo", type=int, default=0, help="MI355_GEMM_ALGO_* override for the headline GEMM")
    ap.add_argument("--dist", choices=("native", "torch"), default="native",
                    help="N > 1: barrier / max-over-ranks through the library's own RCCL communicator (native: one RCCL, one HIP "
                         "runtime in the process; the unique id travels through the launcher's TCP store) or through torch.distributed")
    ap.add_argument("--extras", default="all", help="comma list of extra sections to run (default: all)")
    ap.add_argument("--reduce-elements", type=int, default=1 << 28, help="elements of the array-wide reduction extra (2^28 f32 = 1 GiB = config C4; rehearsals shrink it)")
    ap.add_argument("--threads", type=int, default=0,
                    help="the reference's process model instead of one process per GPU: ONE process, one host thread + one context per device "
                         "(crates/cubecl-common/src/device/handle/channel.rs:24-37), N devices; headline + the C4 exchange only")
    return ap.parse_args()


class Events:
    """Two HIP events on the library's compute stream (the stream the kernels are launched on)."""

    def __init__(self, client):
        self.c = client
        self.a, self.b = C.c_void_p(), C.c_void_p()
        client._s.check(client.lib.mi355_event_create(client.ctx, C.byref(self.a)))
        client._s.check(client.lib.mi355_event_create(client.ctx, C.byref(self.b)))

    def start(self):
        self.c._s.check(self.c.lib.mi355_event_record(self.c.ctx, self.a, None))

    def stop_ms(self) -> float:
        c = self.c
        c._s.check(c.lib.mi355_event_record(c.ctx, self.b, None))
        c._s.check(c.lib.mi355_event_sync(c.ctx, self.b))
        ms = C.c_float()
        c._s.check(c.lib.mi355_event_elapsed_ms(c.ctx, self.a, self.b, C.byref(ms)))
        return float(ms.value)


def time_op(client, ev, fn, iters, warmup=3):
    """Median-free simple protocol for the extras: `warmup` untimed + `iters` timed launches
    between two events; returns average ms per launch."""
    for _ in range(warmup):
        fn()
    client.sync()
    ev.start()
    for _ in range(iters):
        fn()
    return ev.stop_ms() / iters


def samples_op(client, ev, fn, samples=15, warmup=5):
    """The reference's Benchmark protocol (crates/cubecl-common/src/benchmark.rs:183,268): 5 warm-ups,
    15 samples, sync on both sides of each sample; returns (median_ms, min_ms)."""
    for _ in range(warmup):
        fn()
    client.sync()
    out = []
    for _ in range(samples):
        ev.start()
        fn()
        out.append(ev.stop_ms())
    out.sort()
    return out[len(out) // 2], out[0]


def mapped_libraries():
    """Which copies of the HIP runtime and of RCCL this process has mapped (/proc/self/maps): torch bundles its own
    libamdhip64 / librccl next to /opt/rocm's, and by soname whichever is mapped first serves everybody (review of round 3, weak #10)."""
    out = {}
    try:
        for line in open("/proc/self/maps"):
            path = line.rsplit(" ", 1)[-1].strip()
            base = path.rsplit("/", 1)[-1]
            for key in ("libamdhip64", "librccl", "libmi355cube"):
                if base.startswith(key) and path not in out.setdefault(key, []):
                    out[key].append(path)
    except OSError:
        pass
    return {k: (v[0] if len(v) == 1 else v) for k, v in out.items()}


class Job:
    """barrier + reductions of a few host numbers over the ranks of this launch.  world == 1: nothing.  `native`: the library's
    own communicator (cubecl_amd.sharded.RcclJob); `torch`: torch.distributed (nccl = RCCL, or gloo in rehearsals)."""

    def __init__(self, world):
        self.world, self.kind, self.native, self.dist, self.device, self.why = world, "single", None, None, None, None

    def barrier(self):
        if self.native is not None:
            self.native.barrier()
        elif self.dist is not None:
            if self.device is not None:
                self.dist.barrier(device_ids=[self.device])
            else:
                self.dist.barrier()

    def max_over_ranks(self, values):
        if self.native is not None:
            return self.native.max_over_ranks(values)
        if self.dist is not None:
            import torch
            t = torch.tensor(list(values), dtype=torch.float64, device="cuda" if self.device is not None else "cpu")
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            return [float(v) for v in t]
        return list(values)


def launcher_store(rank, world):
    """The launcher's rendezvous store as a plain key-value store (no process group): torch.distributed.run hosts a TCPStore on
    MASTER_ADDR:MASTER_PORT (TORCHELASTIC_USE_AGENT_STORE=True) and workers connect as clients; without an agent rank 0 hosts it."""
    import datetime
    import torch.distributed as dist
    addr, port = os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ["MASTER_PORT"])
    agent = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "") == "True"
    store = dist.TCPStore(addr, port, world, is_master=(rank == 0 and not agent), timeout=datetime.timedelta(seconds=300),
                          wait_for_workers=False)
    return dist.PrefixStore(f"mi355bench/{os.environ.get('TORCHELASTIC_RUN_ID', 'job')}/{os.environ.get('TORCHELASTIC_RESTART_COUNT', '0')}", store)


HUNG = []          # names of watchdogged sections that did not return (see run_with_watchdog)
THREAD_DEVICE = None    # this rank's device index, for the watchdog's helper threads (set in main)


def run_with_watchdog(fn, seconds):
    """Runs fn() in a helper thread; returns None on success, else a short error string.  ctypes and torch release the GIL
    inside their calls, so a collective that never completes leaves this thread free to give up on it."""
    import threading
    box = {}

    def body():
        try:
            if THREAD_DEVICE is not None:         # torch's current device is per thread: a new thread starts on device 0
                import torch
                torch.cuda.set_device(THREAD_DEVICE)
            fn()
            box["ok"] = True
        except Exception as exc:  # noqa: BLE001
            box["err"] = f"{type(exc).__name__}: {exc}"[:300]
    t = threading.Thread(target=body, daemon=True)
    t.start()
    t.join(seconds)
    if t.is_alive():
        HUNG.append(getattr(fn, "__name__", "section"))
        return f"no completion within {seconds:.0f} s"
    return box.get("err")


def usable_cores():
    """Host cores this process may actually use: the affinity mask, cut by a cgroup CPU quota when the container has one
    (os.cpu_count() alone reports the machine's logical CPUs, whatever the container is allowed)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:                                                        # cgroup v2
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(period)
    except Exception:
        try:                                                    # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and period > 0:
                quota = q / period
        except Exception:
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.999)))
    return n


# Sources a PMC pass is tied to: the committed counters describe ONE binary, so every profiles/pmc_*.json entry carries the
# sha256 of the kernel's sources at the time of the pass (tools/pmc_all.sh) and bench.py prints the figure only when the
# sources it is running from still hash to the same value (and the demangled kernel name is the one expected).
KERNEL_SOURCES = {
    "gemm": ("cubecl_amd/csrc/gemm_lp256qm.hip", "cubecl_amd/csrc/gemm_common.hpp", "cubecl_amd/csrc/internal.hpp"),
    "gemm_q": ("cubecl_amd/csrc/gemm_lp256qm.hip", "cubecl_amd/csrc/gemm_common.hpp", "cubecl_amd/csrc/internal.hpp"),
    "reduce": ("cubecl_amd/csrc/reduce.hip", "cubecl_amd/csrc/internal.hpp"),
}
PROFILES_DIR = ROOT / "profiles"
HEADLINE_KERNEL = "gemm_lp256qm_kernel<1, 1, false>"   # <bf16, one dripped store per K-tile, [N][K] rhs>: bf16 x bf16 -> bf16 C, [N][K] B, persistent, on v_mfma_f32_16x16x32
                                                   # (what rocprofv3 prints; round 5: gemm_lp256m16_kernel<1, 1>; until round 4: gemm_lp256w4_kernel<...>)
HEADLINE_ALGO = 15                                 # MI355_GEMM_ALGO_LP_256QM: what AUTO takes for config C3 (round 6)
REDUCE_SUM_KERNEL = "reduce_kernel<0, 0, 0>"      # <VOP = MI355_REDUCE_SUM, AOP = none, DT = f32> (until round 3: <true, false, 0>)
C5_KERNEL = "gemm_lp256qm_kernel<1, 1, false>"         # the same instantiation on batch 512 x 2048^3 (32 K-tiles per tile; round 5: gemm_lp256q_kernel<1, 1, false>)
C5_ALGO = 15


def kernel_source_sha(kind):
    import hashlib
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES[kind]:
        h.update(rel.encode() + b"\0" + (ROOT / rel).read_bytes())
    return h.hexdigest()[:16]


def _pmc_entry(file, key, kind, kernel):
    """-> (entry, None) when profiles/<file> holds a pass for `key` taken on THIS source tree and kernel, else (None, why)."""
    try:
        ent = json.loads((PROFILES_DIR / file).read_text()).get(key)
    except Exception as exc:
        return None, f"unreadable: {exc}"[:120]
    if not ent:
        return None, "no PMC pass for this size / kernel"
    if kernel not in ent.get("kernel", ""):
        return None, f"stale: pass taken on kernel '{ent.get('kernel', '?')[:80]}', current kernel is '{kernel}'"
    if ent.get("source_sha") != kernel_source_sha(kind):
        return None, f"stale: kernel sources changed since the pass (pass {ent.get('source_sha')}, now {kernel_source_sha(kind)})"
    return ent, None


def _pmc_source(ent):
    return {k: ent.get(k) for k in ("kernel", "git_sha", "source_sha", "date", "tool") if ent.get(k) is not None}


def pmc_traffic(size, algo):
    """HBM bytes per launch of the headline kernel from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json,
    written by tools/pmc_all.sh: separate --pmc passes for FETCH_SIZE and WRITE_SIZE, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950).  None when no pass exists for this size / kernel or when the pass was
    taken on other sources (see pmc_traffic_entry for the reason)."""
    ent, _ = pmc_traffic_entry(size, algo)
    return ent["hbm_bytes_per_launch"] if ent else None


def pmc_traffic_entry(size, algo):
    return _pmc_entry("pmc_traffic.json", f"gemm_bf16_{size}_algo{algo}", "gemm", HEADLINE_KERNEL)


def pmc_mfma_util(size):
    """Matrix-pipe utilisation of the headline kernel from the committed rocprofv3 PMC pass (profiles/pmc_mfma_util.json:
    SQ_VALU_MFMA_BUSY_CYCLES per SIMD over GRBM_GUI_ACTIVE per XCD).  None without a pass on these sources."""
    ent, _ = pmc_mfma_util_entry(size)
    return ent["mfma_util"] if ent else None


def pmc_mfma_util_entry(size):
    return _pmc_entry("pmc_mfma_util.json", f"gemm_bf16_{size}", "gemm", HEADLINE_KERNEL)


ALGO_NAMES = {1: "generic", 2: "f32_mfma", 3: "lp128", 4: "lp256 (alias of lp256w4)", 5: "lp256w4", 6: "lp256p", 7: "lp256q", 8: "skinny", 9: "stream64",
              10: "lp256x128", 11: "nnrows", 12: "lp256x192", 13: "lp192x192", 14: "lp256m16", 15: "lp256qm"}


def gemm_desc(N, m, n, k, dtype_ab, dtype_c, trans_b=1, batch=1, algo=0):
    return N.GemmDesc(m=m, n=n, k=k, batch=batch, lda=k, ldb=(k if trans_b else n), ldc=n, stride_a=m * k,
                      stride_b=n * k, stride_c=m * n, dtype_ab=dtype_ab, dtype_c=dtype_c, trans_a=0, trans_b=trans_b,
                      algo=algo)


def self_spawn(gpus):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: become the launcher.  Re-runs this very command line
    as N ranks under torch.distributed.run (one process per GPU, 127.0.0.1 rendezvous on a free port), passes rank 0's JSON
    line through and leaves with the job's exit code -- so the plain form is a real N-rank job, never a silent 1-rank one."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ, BENCH_SELF_SPAWNED="1")
    env.setdefault("OMP_NUM_THREADS", "1")
    sys.exit(subprocess.call(cmd, env=env))


def main_threads(args):
    """`bench.py --threads N`: the reference's own multi-device model -- one process, one server per DeviceId, each driven by its
    own host thread (DeviceHandle: crates/cubecl-common/src/device/handle/channel.rs:24-37), one communicator per sorted id set
    joined from the device threads (comm_init blocks until every rank has called it: crates/cubecl-cuda/src/compute/
    server.rs:669-703).  Same step, same timing rule as the one-process-per-GPU form: every thread warms up, all meet at a host
    barrier behind their own device sync, every thread launches K steps and waits for its stream, the SLOWEST thread's wall
    time counts, value = N x work / that.  Then config C4 end to end: each device reduces its 1/N slice (fused sum + argmax)
    and runs the one-collective exchange.  One JSON line, n_gpus = N, config.process_model says which model ran."""
    import threading

    import numpy as np

    from cubecl_amd import DeviceId, ElemType, Mi355Runtime, TensorHandle, sharded
    from cubecl_amd import _native as N
    n, S, K, W = args.threads, args.size, args.steps, args.warmup
    ids = [DeviceId(0, i) for i in range(n)]
    uid = bytes(Mi355Runtime.client(ids[0]).comm_unique_id())
    gate = threading.Barrier(n)
    out, errors = [None] * n, []
    flop = 2.0 * S * S * S
    n_total = args.reduce_elements

    def device_thread(i):
        try:
            c = Mi355Runtime.client(ids[i])
            lib, ctx = c.lib, c.ctx
            c.comm_init(ids, uid, rank=i)
            a = TensorHandle.uniform(c, (S, S), ElemType.BF16, SEED, 100 + i, -1.0, 1.0)
            b = TensorHandle.uniform(c, (S, S), ElemType.BF16, SEED, 200 + i, -1.0, 1.0)
            cc = c.empty(S * S * 2)
            d = gemm_desc(N, S, S, S, N.DTYPE_BF16, N.DTYPE_BF16, trans_b=1, algo=args.algo)
            sel = C.c_int32(args.algo)
            if args.algo == 0:
                c._s.check(lib.mi355_gemm_select(ctx, C.byref(d), C.byref(sel)))
            pa, pb, pc = C.c_void_p(a.device_ptr()), C.c_void_p(b.device_ptr()), C.c_void_p(cc.device_ptr())
            step = lambda: c._s.check(lib.mi355_gemm(ctx, None, C.byref(d), pa, pb, pc))
            for _ in range(W):
                step()
            c.sync()
            gate.wait()
            t0 = time.perf_counter()
            for _ in range(K):
                step()
            c.sync()
            dt = time.perf_counter() - t0
            gate.wait()
            # config C4: this device's slice, then the exchange (one all-gather + the combine kernel), 20 steps
            start, count = sharded.shard_aligned_range(n_total, i, n, 4)
            x = TensorHandle.uniform(c, (max(count, 4),), ElemType.F32, SEED, 300 + i, 0.0, 1.0)
            ws, outs = c.empty(1 << 17), c.empty(64)
            p_val, p_sum, p_idx = (C.c_void_p(outs.device_ptr() + o) for o in (0, 4, 8))
            rec = outs.offset_end_by(outs.size - 16)
            g_sum, g_val, g_idx = (outs.offset_start_by(o).offset_end_by(outs.size - o - w) for o, w in ((32, 4), (36, 4), (40, 8)))
            ex = sharded.RcclExchange(c, ids, i)
            starts = [sharded.shard_aligned_range(n_total, r, n, 4)[0] for r in range(n)]

            def c4():
                c._s.check(lib.mi355_sum_argmax_f32(ctx, None, C.c_void_p(x.device_ptr()), count, p_sum, p_val, p_idx, C.c_void_p(ws.device_ptr()), ws.size))
                ex.exchange_on_device(rec, starts, g_sum, g_val, g_idx)
            for _ in range(3):
                c4()
            c.sync()
            gate.wait()
            t0 = time.perf_counter()
            for _ in range(20):
                c4()
            c.sync()
            dt4 = (time.perf_counter() - t0) / 20
            gate.wait()
            got = np.frombuffer(c.read_one(outs), dtype=np.uint8)
            out[i] = {"seconds": dt, "algo": sel.value, "c4_seconds": dt4, "c4_sum": float(got[32:36].view(np.float32)[0]),
                      "c4_max": float(got[36:40].view(np.float32)[0]), "c4_index": int(got[40:48].view(np.uint64)[0]),
                      "device": c.properties().name.decode()}
        except BaseException as exc:  # noqa: BLE001
            errors.append(f"device {i}: {type(exc).__name__}: {exc}"[:300])
            gate.abort()
    threads = [threading.Thread(target=device_thread, args=(i,), daemon=True) for i in range(n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(600)
    if errors or any(t.is_alive() for t in threads) or any(o is None for o in out):
        sys.stderr.write("bench.py --threads: " + "; ".join(errors or ["a device thread did not finish"]) + "\n")
        sys.stderr.flush()
        os._exit(3)
    worst = max(o["seconds"] for o in out)
    worst4 = max(o["c4_seconds"] for o in out)
    agree = len({(o["c4_sum"], o["c4_max"], o["c4_index"]) for o in out}) == 1
    value = n * flop * K / worst / 1e12
    print(json.dumps({
        "metric": "GEMM TFLOP/s (8192^3 bf16) + reduce GB/s vs roofline", "value": round(value, 2), "unit": "TFLOP/s", "n_gpus": n, "steps": K, "warmup": W,
        "ms_per_step": round(worst * 1e3 / K, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"{S}x{S}x{S} bf16 GEMM, f32 accumulate, bf16 C (BASELINE config C3), one per GPU",
                   "process_model": "one process, one host thread + one mi355_ctx per device (the reference's DeviceHandle model); one communicator "
                                    "joined from the device threads",
                   "kernel": ALGO_NAMES.get(out[0]["algo"], str(out[0]["algo"])), "parallelism": f"batch-sharded x{n}, no data-path collective"},
        "roofline": {"bound": "mfma", "achieved": round(value / n, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(value / n / PEAK_BF16_TFLOPS, 4),
                     "traffic": None, "reduce_sum_argmax_exchange_ms": round(worst4 * 1e3, 4),
                     "reduce_sum_argmax_exchange_GBs_whole_job": round(n_total * 4 / worst4 / 1e9, 1)},
        "extra": {"reduce_1GiB_f32": {"sharded_sum_argmax_exchange": {"ms": round(worst4 * 1e3, 4), "sum": out[0]["c4_sum"], "argmax_value": out[0]["c4_max"],
                                                                       "argmax_index": out[0]["c4_index"], "every_device_holds_the_same_result": agree,
                                                                       "exchange": "ONE RCCL all-gather (16 B per rank: max, partial sum, index) + combine kernel"}}},
        "device": out[0]["device"]}), flush=True)


def main():
    args = parse_args()
    if args.threads > 0:
        return main_threads(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_spawn(args.gpus)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        # the launcher decides how many ranks exist; a line claiming another count would be a lie
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    args.gpus = world

    # N = 1 never imports torch (review of round 5, weak #4): whichever libamdhip64 is mapped first serves the whole process by
    # soname, and torch bundles its own (HIP 7.0) next to the /opt/rocm runtime (7.2) the library is built and tested against --
    # the timings then came from another HIP stack than the parity evidence.  Device presence and the synchronisation of the
    # contract's bracket come from the library (mi355_device_count; client.sync() = mi355_sync on the stream every kernel of this
    # file is launched on); torch is loaded for N > 1 only, where the launcher's TCP store and the fallback process group are its.
    torch = dist = None
    if world > 1:
        import torch
        import torch.distributed as dist

    # BENCH_NO_TORCH_CUDA=1 (CPU test tier only, tests/test_bench_cpu.py): MI355CUBE_LIB then points at a test build of the host
    # runtime with stand-in kernels, so that the N > 1 control flow of this file -- rendezvous, native barrier / max-over-ranks,
    # the C4 exchange -- runs to completion on a box without a device.  Never set by the driver; such a run's figures mean nothing.
    fake = os.environ.get("BENCH_NO_TORCH_CUDA") == "1"
    from cubecl_amd import _native as _N0
    _lib0 = _N0.load()
    _count = C.c_int32(0)
    visible = int(_count.value) if _lib0.mi355_device_count(C.byref(_count)) == _N0.OK else 0
    if not fake and visible < 1:
        raise SystemExit("bench.py needs a GPU (MI355X); none visible")
    # Rehearsal hooks (single-GPU pod only, never set by the driver): BENCH_FORCE_DEVICE puts every rank on one device and
    # BENCH_DIST_BACKEND=gloo replaces RCCL for torch's own collectives, so that the N > 1 control flow of this file can be
    # exercised where only one GPU exists (the RCCL exchange of the extras then fails cleanly: two ranks on one device).
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    dev_index = local_rank                      # the library's device (DeviceId.index_id)
    if "BENCH_FORCE_DEVICE" in os.environ:
        local_rank = int(os.environ["BENCH_FORCE_DEVICE"])
        dev_index = local_rank
    elif not fake and visible < world:
        raise SystemExit(f"bench.py: {world} ranks asked for, {visible} GPUs visible")
    if not fake and torch is not None:
        torch.cuda.set_device(local_rank)
    global THREAD_DEVICE
    THREAD_DEVICE = None if (fake or torch is None) else local_rank

    def device_sync():
        # the contract's "torch.cuda.synchronize()".  N > 1: torch's own; N = 1: the library's stream synchronisation (all of this
        # file's device work is queued on that stream) -- the same guarantee without a second HIP runtime in the process
        if fake:
            return
        if torch is not None:
            torch.cuda.synchronize()
        else:
            client.sync()

    from cubecl_amd import DeviceId, ElemType, Mi355Runtime, TensorHandle, ops
    from cubecl_amd import _native as N

    client = Mi355Runtime.client(DeviceId(0, dev_index))
    lib, ctx = client.lib, client.ctx
    props = client.properties()
    ev = Events(client)

    job = Job(world)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        store = launcher_store(rank, world)
        native_ok, why = False, None
        if args.dist == "native":
            # The library's own communicator carries the job-level barrier and the max over ranks: no torch process group, so one
            # RCCL communicator and one librccl in the process.  comm_init is collective; every rank then says through the store
            # whether it came up, and ALL ranks fall back to torch.distributed together if any did not.
            from cubecl_amd import sharded

            def native_init():
                job.native = sharded.RcclJob(client, [DeviceId(0, i) for i in range(world)] if "BENCH_FORCE_DEVICE" not in os.environ
                                             else [DeviceId(0, dev_index)] * world, rank, store)
                job.native.barrier()
            why = run_with_watchdog(native_init, 240.0)
            if HUNG:                    # stuck inside ncclCommInitRank with the context locked: nothing on this context can run any more
                sys.stderr.write(f"bench.py rank {rank}: native RCCL rendezvous did not complete ({why}); use --dist torch\n")
                sys.stderr.flush()
                os._exit(3)
            store.set(f"native_ok/{rank}", b"1" if why is None else b"0")
            native_ok = all(bytes(store.get(f"native_ok/{r}")) == b"1" for r in range(world))
            if not native_ok:
                job.native = None
        if native_ok:
            job.kind = "native"
        else:
            job.why = why or ("--dist torch" if args.dist == "torch" else "another rank's native rendezvous failed")
            if backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
                job.device = local_rank
            else:
                dist.init_process_group(backend=backend)
            job.dist, job.kind = dist, f"torch:{backend}"

    barrier = job.barrier

    def job_seconds(fn, iters, warmup=2):
        """Job-level time of one `fn` (the protocol of the headline, applied to an extra): every rank warms up, barrier,
        every rank launches `iters` x fn back to back and waits for its own stream, and the slowest rank's wall time
        counts -- so a whole-job rate is units of ALL ranks / this time, never rank 0's own figure times N."""
        for _ in range(warmup):
            fn()
        client.sync()
        barrier()
        device_sync()
        t1 = time.perf_counter()
        for _ in range(iters):
            fn()
        client.sync()
        device_sync()
        dt = (time.perf_counter() - t1) / iters
        barrier()
        return job.max_over_ranks([dt])[0]

    # ------------------------------------------------------------------ headline: C3 ------------------
    S = args.size
    a = TensorHandle.uniform(client, (S, S), ElemType.BF16, SEED, 100 + rank, -1.0, 1.0)
    b = TensorHandle.uniform(client, (S, S), ElemType.BF16, SEED, 200 + rank, -1.0, 1.0)   # stored [N][K]
    c = TensorHandle.new_contiguous((S, S), client.empty(S * S * 2), ElemType.BF16)
    desc = gemm_desc(N, S, S, S, N.DTYPE_BF16, N.DTYPE_BF16, trans_b=1, algo=args.algo)
    sel = C.c_int32(args.algo)
    if args.algo == 0:
        client._s.check(lib.mi355_gemm_select(ctx, C.byref(desc), C.byref(sel)))
    pa, pb, pc = C.c_void_p(a.device_ptr()), C.c_void_p(b.device_ptr()), C.c_void_p(c.device_ptr())

    def step():
        client._s.check(lib.mi355_gemm(ctx, None, C.byref(desc), pa, pb, pc))

    clk = client.empty(2 * 8192)     # two samples of {shader ticks, 100 MHz ticks} per CU bracketing the timed region
    lib.mi355_memset(ctx, None, C.c_void_p(clk.device_ptr()), 0, 2 * 8192)
    p_clk0, p_clk1 = C.c_void_p(clk.device_ptr()), C.c_void_p(clk.device_ptr() + 8192)
    for _ in range(args.warmup):
        step()
    # the clock probe's first launch loads its code object (0.5 ms on the host, tools/dev/sync_cost.py): not inside the timed region
    lib.mi355_probe_clock(ctx, None, p_clk0)
    # Plateau warm-up, as the reference's ThroughputBenchmarker does before sampling
    # (crates/cubecl-runtime/src/throughput/benchmarker.rs:40-143: grow the warm-up until the rate stops
    # moving): from idle the chip needs ~25 ms of this kernel before DVFS settles (first 30 launches
    # 1 290 TFLOP/s, every later block 1 425-1 430, flat for 1.5 s -- profiles/r01_power_ablation.md).  Blocks
    # of 20 untimed launches until two consecutive blocks agree within 1.5 %, at most 12 blocks.
    plateau_steps, last = 0, None
    if not args.no_plateau_warmup:
        for _ in range(12):
            ev.start()
            for _ in range(20):
                step()
            cur = ev.stop_ms()
            plateau_steps += 20
            if last is not None and abs(cur - last) <= 0.015 * last:
                break
            last = cur
    client.sync()
    barrier()
    device_sync()                        # torch.cuda.synchronize() (N > 1) / client.sync() (N = 1)
    t0 = time.perf_counter()
    lib.mi355_probe_clock(ctx, None, p_clk0)
    ev.start()
    for _ in range(args.steps):
        step()
    kernel_ms = ev.stop_ms()
    lib.mi355_probe_clock(ctx, None, p_clk1)
    device_sync()                        # torch.cuda.synchronize()
    client.sync()
    elapsed_own = time.perf_counter() - t0     # this rank's K steps, launched and drained
    barrier()
    elapsed = time.perf_counter() - t0         # ... and every other rank's (the closing barrier is inside the bracket)
    import numpy as _np
    ticks = _np.frombuffer(client.read_one(clk), dtype=_np.uint64).reshape(2, 512, 2).astype(_np.float64)
    # s_memtime is local to a CU: pair the two samples slot by slot (same CU), median over the CUs seen twice
    ok = (ticks[0, :, 1] > 0) & (ticks[1, :, 1] > ticks[0, :, 1]) & (ticks[1, :, 0] > ticks[0, :, 0])
    per_cu = (ticks[1, ok, 0] - ticks[0, ok, 0]) / (ticks[1, ok, 1] - ticks[0, ok, 1]) * 0.1      # 100 MHz reference
    eff_clock_ghz = float(_np.median(per_cu)) if per_cu.size else float("nan")
    if world > 1:
        elapsed, kernel_ms, elapsed_own = job.max_over_ranks([elapsed, kernel_ms, elapsed_own])
    flop = 2.0 * S * S * S
    ms_per_step = elapsed * 1e3 / args.steps
    value = world * flop * args.steps / elapsed / 1e12
    # roofline.rs:76-92 `score_resources`: one measured duration against the resource that binds this kernel
    from cubecl_amd.roofline import ResourceBound, score_resources
    gemm_score = score_resources(kernel_ms / args.steps * 1e-3, [ResourceBound(int(flop), PEAK_BF16_TFLOPS * 1e12)])[0]
    achieved = gemm_score.achieved_per_s / 1e12

    tr_ent, tr_why = pmc_traffic_entry(S, sel.value) if sel.value == HEADLINE_ALGO else (None, "no PMC pass for this kernel")
    mu_ent, mu_why = pmc_mfma_util_entry(S) if sel.value == HEADLINE_ALGO else (None, "no PMC pass for this kernel")
    result = {
        "metric": "GEMM TFLOP/s (8192^3 bf16) + reduce GB/s vs roofline",
        "value": round(value, 2), "unit": "TFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"{S}x{S}x{S} bf16 GEMM, f32 accumulate, bf16 C (BASELINE config C3), one per GPU",
                   "layout": "A[M,K] row-major; B stored [N][K] (Out = Lhs*Rhs^T, the cmma tests' ColMajor-B form)",
                   "operands": "uniform[-1,1) counter RNG seed 0x5EEDC0BE, generated in HBM",
                   "kernel": ALGO_NAMES.get(sel.value, str(sel.value)),
                   "parallelism": f"batch-sharded x{world}, no data-path collective",
                   "plateau_warmup_steps": plateau_steps,
                   "job_collectives": job.kind + (f" (native refused: {job.why})"[:160] if job.why and args.dist == "native" else ""),
                   "ms_per_step_before_closing_barrier": round(elapsed_own * 1e3 / args.steps, 4)},
        # Key ORDER is part of the record: the driver keeps the first ~20 keys of this object, cut at 40 characters (review of round 5,
        # weak #8) -- so the reduce half of the metric (C4), C5 and C2 sit right behind the GEMM's six, as short flat scalars, filled
        # in by the extras further down (None = that extra did not run); everything else follows.
        "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(gemm_score.fraction_of_peak, 4), "traffic": (tr_ent or {}).get("hbm_bytes_per_launch"),
                     "reduce_sum_achieved_GBs": None, "reduce_sum_frac": None, "reduce_sum_traffic": None, "reduce_sum_kernel_ms": None,
                     "reduce_shard8_sum_us": None, "reduce_shard8_sum_frac": None, "reduce_shard8_fused_us": None, "reduce_shard8_fused_frac": None,
                     "c4_8gpu_projected_us": None, "c5_whole_job_TFLOPs": None, "c5_frac": None, "c2_f32_frac": None,
                     "kernel_ms": round(kernel_ms / args.steps, 4), "shader_clock_GHz": round(eff_clock_ghz, 3),
                     "frac_of_peak_at_clock": round(achieved / (PEAK_BF16_TFLOPS * eff_clock_ghz / 2.4), 4),
                     "mfma_util_pmc": (mu_ent or {}).get("mfma_util"),
                     "traffic_source": _pmc_source(tr_ent) if tr_ent else tr_why,
                     "algorithmic_bytes_per_launch": 3 * S * S * 2,
                     "flop_per_launch": flop,
                     "mfma_util_source": _pmc_source(mu_ent) if mu_ent else mu_why,
                     # the chip clocks down to its power budget on random operands (MI355X_MICROARCH.md "DVFS
                     # give-back"): the 2.5 PFLOP/s peak assumes 2.4 GHz; these two lines price the kernel
                     # against the matrix-pipe rate at the clock it actually ran at
                     # (shader_clock_GHz, frac_of_peak_at_clock: among the first keys above)
                     # shader cycles of one launch: what stays put between boxes whose sustained clocks differ (a rocprof summary taken
                     # on one box reconciles with a bench line of another through this product, review of round 4, next #4)
                     "kernel_ms_times_shader_clock_GHz": round(kernel_ms / args.steps * eff_clock_ghz, 4)},
        "device": props.name.decode() + " " + props.gcn_arch_name.decode(),
    }

    # the same launch under the reference's Benchmark protocol (crates/cubecl-common/src/benchmark.rs:160-300: sync around every
    # sample), outside the timed region: the fixed name every config's second figure carries
    # (with the extras only: `--no-extras` is the command the rocprofv3 summary profiles, whose per-kernel average should be that of
    # the warm-up + timed launches and nothing else)
    if not args.no_extras:
        med_ms, _ = samples_op(client, ev, step)
        result["roofline"]["frac_per_sample_median"] = round(flop / med_ms / 1e9 / PEAK_BF16_TFLOPS, 4)
    libs = mapped_libraries()
    for key in ("libamdhip64", "librccl"):          # which copies serve this process (flat strings: the driver's record keeps scalars)
        result["config"][key] = str(libs.get(key, "not mapped"))
    result["config"]["torch_in_process"] = "torch" in sys.modules     # False at N = 1: one HIP runtime, the one the library is tested against
    extra, errors = {}, {}
    wanted = None if args.extras == "all" else set(args.extras.split(","))

    def guarded(name, fn):
        if wanted is not None and name not in wanted:
            return
        try:
            extra[name] = fn()
        except Exception as exc:  # keep the headline line even if an extra fails
            errors[name] = f"{type(exc).__name__}: {exc}"[:300]

    # ------------------------------------------------------------------ CPU baseline ---------------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:       # N = 1 only: at N > 1 it would load the cores the other ranks launch from
        def cpu_baseline():
            import numpy as np
            import oracle
            cores = usable_cores()
            size = 1024
            secs = None
            while True:
                x = oracle.fill_uniform(size * size, 100, -1.0, 1.0)
                y = oracle.fill_uniform(size * size, 200, -1.0, 1.0)
                s, _ = oracle.cpu_gemm(oracle.to_bf16(x), oracle.to_bf16(y), size, size, size, dtype_ab=oracle.DT_BF16,
                                       dtype_c=oracle.DT_BF16, trans_b=True, units=cores)
                secs = s
                if s > 4.0 or size >= S:          # up to the benched problem itself when the cores finish it within the budget
                    break
                size *= 2
            return {"value": round(2.0 * size ** 3 / secs / 1e12, 5), "unit": "TFLOP/s", "cores": cores, "kind": "port",
                    "sample": f"{size}^3 bf16 GEMM (same RNG/layout), {secs:.2f} s, oracle_cpu_gemm: one worker per "
                              "cube unit as cubecl-cpu schedules (threadpool/mod.rs:80-99); "
                              f"{os.cpu_count()} logical CPUs on the host, {cores} usable by this process"}
        try:
            result["cpu_baseline"] = cpu_baseline()
        except Exception as exc:
            result["cpu_baseline"] = {"value": None, "unit": "TFLOP/s", "cores": os.cpu_count(), "kind": "port",
                                      "sample": f"failed: {exc}"[:200]}

        def parity_of_this_run():
            # The tolerance, stated where the number is printed (review of round 4, weak #1): north_star says "within 1e-5 relative
            # for f32"; every parity test reads that against sum_k |a||b| (norm-wise), because a dot product of 8192 mixed-sign terms
            # cancels.  Here, on THIS run's operands and kernel (f32 C, four sampled rows x all columns): the device's error both
            # ways, beside the reference's OWN arithmetic -- operands widened to f32, `sum += l * r` sequentially, separate multiply
            # and add (crates/cubecl-core/src/runtime_tests/cmma.rs:695-722; oracle_gemm, part of the CPU leg) -- against an f64
            # product.  The literal reading (|err| / |ref|) is missed by that loop too, by the same factor.
            import numpy as np
            import oracle
            rows = np.array([5003 % S, 0, S - 1, 257])
            c32 = client.empty(S * S * 4)
            d32 = gemm_desc(N, S, S, S, N.DTYPE_BF16, N.DTYPE_F32, trans_b=1, algo=args.algo)
            client._s.check(lib.mi355_gemm(ctx, None, C.byref(d32), pa, pb, C.c_void_p(c32.device_ptr())))
            got = np.stack([np.frombuffer(client.read_one(c32.offset_start_by(int(r) * S * 4).offset_end_by((S - 1 - int(r)) * S * 4)), dtype=np.float32)
                            for r in rows]).astype(np.float64)
            a_bits = oracle.to_bf16(oracle.fill_uniform(S * S, 100, -1.0, 1.0)).reshape(S, S)[rows]
            b_bits = oracle.to_bf16(oracle.fill_uniform(S * S, 200, -1.0, 1.0))
            loop = oracle.gemm(np.ascontiguousarray(a_bits), b_bits, len(rows), S, S, dtype_ab=oracle.DT_BF16, dtype_c=oracle.DT_F32,
                               trans_b=True).reshape(len(rows), S).astype(np.float64)
            A = oracle.from_bf16(np.ascontiguousarray(a_bits)).reshape(len(rows), S).astype(np.float64)
            ref, bound = np.empty((len(rows), S)), np.empty((len(rows), S))
            for j0 in range(0, S, 1024):                                  # B in column panels: 64 MiB of f64 at a time
                Bp = oracle.from_bf16(b_bits.reshape(S, S)[j0:j0 + 1024]).reshape(-1, S).astype(np.float64).T
                ref[:, j0:j0 + 1024], bound[:, j0:j0 + 1024] = A @ Bp, np.abs(A) @ np.abs(Bp)
            e_dev, e_loop = np.abs(got - ref), np.abs(loop - ref)
            solid = np.abs(ref) >= 1e-3 * bound                           # outputs that are not themselves cancellation residue
            return {"reading": "|err| <= 1e-5 * sum_k |a||b| (norm-wise); the literal |err| / |ref| is printed beside it",
                    "sample": f"rows {rows.tolist()} x all {S} columns of this run's operands, f32 C, against an f64 product",
                    "device_max_err_over_sum_abs_products": float((e_dev / bound).max()),
                    "reference_loop_max_err_over_sum_abs_products": float((e_loop / bound).max()),
                    "literal_max_rel": float((e_dev[solid] / np.abs(ref[solid])).max()),
                    "reference_loop_literal_max_rel": float((e_loop[solid] / np.abs(ref[solid])).max()),
                    "within_tolerance": bool(np.all(e_dev <= 1e-5 * bound))}
        if not fake:
            # flat scalars: the driver's record of `config` keeps scalars only (review of round 5, weak #1)
            try:
                par = parity_of_this_run()
                result["config"].update({"parity_reading": par["reading"], "parity_norm_max": par["device_max_err_over_sum_abs_products"],
                                         "parity_literal_max": par["literal_max_rel"],
                                         "parity_ref_loop_norm_max": par["reference_loop_max_err_over_sum_abs_products"],
                                         "parity_ref_loop_literal_max": par["reference_loop_literal_max_rel"],
                                         "parity_within_tolerance": par["within_tolerance"], "parity_sample": par["sample"]})
            except Exception as exc:  # noqa: BLE001
                result["config"]["parity_error"] = f"{type(exc).__name__}: {exc}"[:200]

    # ------------------------------------------------------------------ extras ---------------------------
    if not args.no_extras:
        del c
        sink = client.empty(256)

        def probes():
            buf = client.empty(1 << 30)
            client._s.check(lib.mi355_memset(ctx, None, buf.device_ptr(), 0, 1 << 30))
            ms = time_op(client, ev, lambda: client._s.check(
                lib.mi355_probe_memory_read(ctx, None, buf.device_ptr(), 1 << 30, 1, sink.device_ptr())), 20)
            out = {"hbm_read_GBs": round((1 << 30) / ms / 1e6, 1)}
            for name, dt, peak in (("mfma_bf16_TFLOPs", N.DTYPE_BF16, PEAK_BF16_TFLOPS), ("mfma_f32_TFLOPs", N.DTYPE_F32, PEAK_F32_TFLOPS)):
                n_ops = C.c_uint64()
                iters = 20000 if dt == N.DTYPE_BF16 else 4000
                ms = time_op(client, ev, lambda: client._s.check(
                    lib.mi355_probe_mfma(ctx, None, dt, iters, sink.device_ptr(), C.byref(n_ops))), 5)
                out[name] = round(n_ops.value / ms / 1e9, 1)
            # matrix-pipe ceiling on the benchmark's operand distribution (register-resident, no memory traffic)
            n_ops = C.c_uint64()
            ms = time_op(client, ev, lambda: client._s.check(
                lib.mi355_probe_mfma_data(ctx, None, 1, 20000, sink.device_ptr(), C.byref(n_ops))), 5)
            out["mfma_bf16_uniform_operands_TFLOPs"] = round(n_ops.value / ms / 1e9, 1)
            # fp8 (v_mfma_f32_32x32x64_f8f6f4, e4m3): all-ones operands, then uniform[-1,1) operands
            n_ops = C.c_uint64()
            ms = time_op(client, ev, lambda: client._s.check(
                lib.mi355_probe_mfma(ctx, None, N.DTYPE_F8E4M3, 10000, sink.device_ptr(), C.byref(n_ops))), 5)
            out["mfma_fp8_TFLOPs"] = round(n_ops.value / ms / 1e9, 1)
            ms = time_op(client, ev, lambda: client._s.check(
                lib.mi355_probe_mfma_data(ctx, None, 2, 10000, sink.device_ptr(), C.byref(n_ops))), 5)
            out["mfma_fp8_uniform_operands_TFLOPs"] = round(n_ops.value / ms / 1e9, 1)
            for key, mode in (("mfma_mxfp4_TFLOPs", 3), ("mfma_mxfp4_random_operands_TFLOPs", 4)):
                ms = time_op(client, ev, lambda: client._s.check(
                    lib.mi355_probe_mfma_data(ctx, None, mode, 20000, sink.device_ptr(), C.byref(n_ops))), 5)
                out[key] = round(n_ops.value / ms / 1e9, 1)
            # the reference's remaining throughput probes (examples/throughput: copy, write, compute-direct, launch)
            buf2 = client.empty(1 << 30)
            ms = time_op(client, ev, lambda: client._s.check(
                lib.mi355_probe_memory_copy(ctx, None, buf.device_ptr(), buf2.device_ptr(), 1 << 30)), 20)
            out["hbm_copy_GBs_read_plus_write"] = round(2 * (1 << 30) / ms / 1e6, 1)
            ms = time_op(client, ev, lambda: client._s.check(lib.mi355_probe_memory_write(ctx, None, buf2.device_ptr(), 1 << 30)), 20)
            out["hbm_write_GBs"] = round((1 << 30) / ms / 1e6, 1)
            n_ops = C.c_uint64()
            ms = time_op(client, ev, lambda: client._s.check(
                lib.mi355_probe_compute_direct(ctx, None, 20000, sink.device_ptr(), C.byref(n_ops))), 5)
            out["fma_f32_TFLOPs"] = round(n_ops.value / ms / 1e9, 1)
            ms = time_op(client, ev, lambda: client._s.check(lib.mi355_probe_launch_overhead(ctx, None, 1000, sink.device_ptr())), 3, warmup=1)
            out["launch_overhead_us"] = round(ms, 3)          # ms per 1000 launches = us per launch
            return out

        def operand_sensitivity():
            # Separates issue efficiency from power (review of round 2, weak #5): the SAME headline kernel, same descriptor, on
            # all-zero and all-ones operands -- no toggling in the multipliers, so the chip holds its clock -- next to the
            # benchmark's uniform[-1,1) operands.  If zeros/ones reach ~2 000+ TFLOP/s the K loop's issue schedule is fine and
            # the gap on random data is the power limit; if they stall near the random-data figure, issue slack is hiding there.
            import numpy as _np2
            out = {}
            za = TensorHandle.new_contiguous((S, S), client.empty(S * S * 2), ElemType.BF16)
            zb = TensorHandle.new_contiguous((S, S), client.empty(S * S * 2), ElemType.BF16)
            zc = client.empty(S * S * 2)
            d = gemm_desc(N, S, S, S, N.DTYPE_BF16, N.DTYPE_BF16, trans_b=1, algo=args.algo)
            call = lambda: client._s.check(lib.mi355_gemm(ctx, None, C.byref(d), za.device_ptr(), zb.device_ptr(), zc.device_ptr()))
            for name, setup in (("zeros", lambda h: lib.mi355_memset(ctx, None, C.c_void_p(h.device_ptr()), 0, S * S * 2)),
                                ("ones", lambda h: lib.mi355_fill_uniform(ctx, None, C.c_void_p(h.device_ptr()), int(ElemType.BF16), S * S, SEED, 1, 1.0, 1.0)),
                                ("uniform", lambda h: lib.mi355_fill_uniform(ctx, None, C.c_void_p(h.device_ptr()), int(ElemType.BF16), S * S, SEED, 100 + (h is zb), -1.0, 1.0))):
                client._s.check(setup(za))
                client._s.check(setup(zb))
                time_op(client, ev, call, 60, warmup=0)                    # past the DVFS ramp
                lib.mi355_probe_clock(ctx, None, p_clk0)
                ms = time_op(client, ev, call, 40, warmup=0)
                lib.mi355_probe_clock(ctx, None, p_clk1)
                tk = _np2.frombuffer(client.read_one(clk), dtype=_np2.uint64).reshape(2, 512, 2).astype(_np2.float64)
                good = (tk[0, :, 1] > 0) & (tk[1, :, 1] > tk[0, :, 1]) & (tk[1, :, 0] > tk[0, :, 0])
                ghz = float(_np2.median((tk[1, good, 0] - tk[0, good, 0]) / (tk[1, good, 1] - tk[0, good, 1]) * 0.1)) if good.any() else float("nan")
                tf = flop / ms / 1e9
                out[name] = {"TFLOPs": round(tf, 1), "frac_of_2.5PF": round(tf / PEAK_BF16_TFLOPS, 4), "shader_clock_GHz": round(ghz, 3),
                             "frac_of_peak_at_clock": round(tf / (PEAK_BF16_TFLOPS * ghz / 2.4), 4)}
            out["gemm_bf16_8192_zero_operands_TFLOPs"] = out["zeros"]["TFLOPs"]
            return out

        def reduce_c4():
            n_total = args.reduce_elements         # 2^28 f32 = 1 GiB (config C4)
            n_local = n_total // world
            x = TensorHandle.uniform(client, (n_local,), ElemType.F32, SEED, 300 + rank, 0.0, 1.0)
            ws = client.empty(1 << 17)
            outs = client.empty(64)
            p_in, p_ws = C.c_void_p(x.device_ptr()), C.c_void_p(ws.device_ptr())
            # the 16-byte record of the sharded job at the start of `outs`: {f32 max, f32 (partial) sum, u64 local index}
            p_val, p_sum, p_idx = (C.c_void_p(outs.device_ptr() + o) for o in (0, 4, 8))
            res = {"elements_per_gpu": n_local, "bytes_per_gpu": n_local * 4}
            b2b_ms, med_frac = {}, {}
            for name, fn in (
                ("sum", lambda: lib.mi355_reduce_sum_f32(ctx, None, p_in, n_local, p_sum, p_ws, ws.size)),
                ("argmax", lambda: lib.mi355_argmax_f32(ctx, None, p_in, n_local, p_val, p_idx, p_ws, ws.size)),
                ("sum_argmax_fused", lambda: lib.mi355_sum_argmax_f32(ctx, None, p_in, n_local, p_sum, p_val, p_idx, p_ws, ws.size)),
            ):
                med, best = samples_op(client, ev, lambda: client._s.check(fn()))
                gbs = n_local * 4 / med / 1e6
                b2b = time_op(client, ev, lambda: client._s.check(fn()), iters=20, warmup=2)   # 20 launches, one event pair
                b2b_ms[name] = b2b
                job_s = job_seconds(lambda: client._s.check(fn()), iters=20)      # barrier -> all ranks launch -> sync -> max over ranks
                res[name] = {"median_ms": round(med, 4), "min_ms": round(best, 4), "GBs_per_gpu": round(gbs, 1),
                             "frac_of_8TBs": round(gbs / PEAK_HBM_GBS, 4),
                             "back_to_back_ms": round(b2b, 4), "back_to_back_GBs": round(n_local * 4 / b2b / 1e6, 1),
                             "job_ms": round(job_s * 1e3, 4), "GBs_total": round(n_local * 4 * world / job_s / 1e9, 1)}
                med_frac[name] = gbs / PEAK_HBM_GBS
            # the second half of the metric ("reduce GB/s vs roofline"): same object shape as the headline roofline
            rd_ent, rd_why = _pmc_entry("pmc_traffic.json", "reduce_1GiB_sum", "reduce", REDUCE_SUM_KERNEL)
            tr = rd_ent["fetch_bytes"] if rd_ent else None
            sum_score = score_resources(b2b_ms["sum"] * 1e-3, [ResourceBound(n_local * 4, PEAK_HBM_GBS * 1e9)])[0]
            res["roofline"] = {"bound": "hbm", "achieved": round(sum_score.achieved_per_s / 1e9, 1), "peak": PEAK_HBM_GBS,
                               "unit": "GB/s", "frac": round(sum_score.fraction_of_peak, 4), "traffic": tr if world == 1 else None,
                               "traffic_source": (_pmc_source(rd_ent) if rd_ent else rd_why) if world == 1 else "pass is for the 1-GPU size",
                               "algorithmic_bytes_per_launch": n_local * 4, "kernel_ms": round(b2b_ms["sum"], 4),
                               "frac_per_sample_median": round(med_frac["sum"], 4)}
            # ... and where the driver's record keeps it: flat scalars beside the GEMM's in the top-level roofline object.  `frac` is
            # the claim (average of 20 back-to-back launches between one HIP event pair, the protocol of the GEMM roofline and of the
            # contract's timed region); `frac_per_sample_median` is the reference's Benchmark protocol (a sync around every sample,
            # crates/cubecl-common/src/benchmark.rs:160-300), kept beside it.
            rl = res["roofline"]
            result["roofline"].update({
                "reduce_sum_bound": "hbm", "reduce_sum_bytes_per_gpu": n_local * 4, "reduce_sum_achieved_GBs": rl["achieved"],
                "reduce_sum_peak_GBs": PEAK_HBM_GBS, "reduce_sum_frac": rl["frac"], "reduce_sum_frac_per_sample_median": rl["frac_per_sample_median"],
                "reduce_sum_kernel_ms": rl["kernel_ms"], "reduce_sum_traffic": rl["traffic"],
                "reduce_argmax_frac": round(n_local * 4 / b2b_ms["argmax"] / 1e6 / PEAK_HBM_GBS, 4),
                "reduce_sum_argmax_fused_frac": round(n_local * 4 / b2b_ms["sum_argmax_fused"] / 1e6 / PEAK_HBM_GBS, 4),
                "reduce_sum_GBs_whole_job": res["sum"]["GBs_total"]})
            result["roofline"]["reduce_1GiB_sum"] = {k: rl[k] for k in ("achieved", "peak", "frac", "traffic", "kernel_ms", "frac_per_sample_median")}
            if rank == 0 and world == 1 and not args.no_cpu_baseline:
                # the reduce half's CPU baseline (north_star: "alongside cubecl-cpu timed on the box's own host cores"): the
                # oracle's threaded fused sum + argmax -- one contiguous slice per worker, one worker per usable core, the way
                # cubecl-cpu hands cube units to its pool (threadpool/mod.rs:80-99) -- over the same 1 GiB of the same RNG stream
                try:
                    import numpy as np
                    import oracle
                    cores = usable_cores()
                    hx = oracle.fill_uniform(n_total, 300, 0.0, 1.0)
                    times, spent = [], 0.0
                    while len(times) < 3 or (spent < 6.0 and len(times) < 12):
                        secs, hsum, hidx = oracle.cpu_sum_argmax(hx, cores)
                        times.append(secs)
                        spent += secs
                    times.sort()
                    med_s = times[len(times) // 2]
                    dev = np.frombuffer(client.read_one(outs), dtype=np.uint8)
                    if isinstance(result.get("cpu_baseline"), dict):
                        result["cpu_baseline"].update({"reduce_value": round(n_total * 4 / med_s / 1e9, 2), "reduce_unit": "GB/s", "reduce_cores": cores,
                                                       "reduce_sample": f"the whole 1 GiB f32 array, fused sum + argmax, median of {len(times)} passes"})
                        result["cpu_baseline"]["reduce"] = {"value": round(n_total * 4 / med_s / 1e9, 2), "unit": "GB/s", "cores": cores}
                    res["cpu_baseline"] = {"value": round(n_total * 4 / med_s / 1e9, 2), "unit": "GB/s", "cores": cores, "kind": "port",
                                           "sample": f"the whole 1 GiB f32 array (same counter RNG stream), fused sum + argmax, median of {len(times)} passes "
                                                     f"({med_s * 1e3:.1f} ms), oracle_cpu_sum_argmax_f32: one contiguous slice per worker thread",
                                           "argmax_index_equals_device": bool(hidx == int(dev[8:16].view(np.uint64)[0])),
                                           "sum_rel_diff_vs_device": float(abs(hsum - float(dev[4:8].view(np.float32)[0])) / abs(hsum))}
                    del hx
                except Exception as exc:  # noqa: BLE001
                    res["cpu_baseline"] = {"value": None, "unit": "GB/s", "kind": "port", "sample": f"failed: {exc}"[:200]}
            if world == 1 and n_total % 32 == 0:
                # Config C4 as BASELINE defines it -- the 128 MiB slice ONE of 8 GPUs owns -- on the one GPU there is: the fused
                # sum + argmax pass walks the eight slices of the same 1 GiB array in rotation, so every slice is cold again when
                # its turn comes (1 GiB is four times the 256 MiB infinity cache; the reference's read probe advances its window
                # for the same reason, crates/cubecl-std/src/throughput/runners/memory_read.rs:106-146), back to back between
                # one event pair (the claim) and per sample; then the exchange behind it on a ONE-rank communicator on the real
                # RCCL -- its launch / fence / kernel latency floor, no wire -- and the two together as one step.
                SH = 8
                n_sh = n_total // SH
                p_sl = [C.c_void_p(x.device_ptr() + 4 * n_sh * i) for i in range(SH)]
                turn = [0]

                def shard_pass(fn_name="mi355_sum_argmax_f32"):
                    i = turn[0] = (turn[0] + 1) % SH
                    if fn_name == "mi355_reduce_sum_f32":
                        client._s.check(lib.mi355_reduce_sum_f32(ctx, None, p_sl[i], n_sh, p_sum, p_ws, ws.size))
                    else:
                        client._s.check(lib.mi355_sum_argmax_f32(ctx, None, p_sl[i], n_sh, p_sum, p_val, p_idx, p_ws, ws.size))
                sh = {"shards": SH, "bytes_per_shard": n_sh * 4}
                for key, fname in (("sum", "mi355_reduce_sum_f32"), ("sum_argmax_fused", "mi355_sum_argmax_f32")):
                    b2b = min(time_op(client, ev, lambda: shard_pass(fname), iters=4 * SH, warmup=SH) for _ in range(3))
                    med, best = samples_op(client, ev, lambda: shard_pass(fname), samples=3 * SH, warmup=SH)
                    sh[key] = {"back_to_back_us": round(b2b * 1e3, 2), "GBs": round(n_sh * 4 / b2b / 1e6, 1), "frac_of_8TBs": round(n_sh * 4 / b2b / 1e6 / PEAK_HBM_GBS, 4),
                               "per_sample_median_us": round(med * 1e3, 2), "per_sample_min_us": round(best * 1e3, 2)}
                res["shard_of_8"] = sh
                fused_us = sh["sum_argmax_fused"]["back_to_back_us"]
                result["roofline"].update({"reduce_shard_of_8_bytes": n_sh * 4, "reduce_shard_of_8_sum_us": sh["sum"]["back_to_back_us"],
                                           "reduce_shard_of_8_sum_frac": sh["sum"]["frac_of_8TBs"],
                                           "reduce_shard_of_8_fused_us": fused_us, "reduce_shard_of_8_fused_frac": sh["sum_argmax_fused"]["frac_of_8TBs"],
                                           "reduce_shard_of_8_fused_per_sample_median_us": sh["sum_argmax_fused"]["per_sample_median_us"],
                                           "reduce_shard8_sum_us": sh["sum"]["back_to_back_us"], "reduce_shard8_sum_frac": sh["sum"]["frac_of_8TBs"],
                                           "reduce_shard8_fused_us": fused_us, "reduce_shard8_fused_frac": sh["sum_argmax_fused"]["frac_of_8TBs"]})

                def exchange_floor():
                    from cubecl_amd import sharded
                    ids = [DeviceId(0, dev_index)]
                    client.comm_init(ids, bytes(client.comm_unique_id()), rank=0)
                    ex = sharded.RcclExchange(client, ids, 0)
                    rec = outs.offset_end_by(outs.size - 16)
                    g_sum = outs.offset_start_by(32).offset_end_by(outs.size - 36)
                    g_val = outs.offset_start_by(36).offset_end_by(outs.size - 40)
                    g_idx = outs.offset_start_by(40).offset_end_by(outs.size - 48)
                    out = {}
                    for mode in ("gather", "all_reduce"):
                        t = min(time_op(client, ev, lambda: ex.exchange_on_device(rec, [0], g_sum, g_val, g_idx, mode=mode), iters=20, warmup=3) for _ in range(3))
                        out[mode + "_us"] = round(t * 1e3, 2)

                    def step():
                        shard_pass()
                        ex.exchange_on_device(rec, [0], g_sum, g_val, g_idx)
                    t = min(time_op(client, ev, step, iters=4 * SH, warmup=SH) for _ in range(3))
                    out["shard_pass_plus_exchange_us"] = round(t * 1e3, 2)
                    import numpy as np
                    # after a step the combine of ONE record must reproduce the local pass bit for bit
                    got = np.frombuffer(client.read_one(outs), dtype=np.uint8)
                    out["combine_of_one_record_is_identity"] = bool(bytes(got[0:4]) == bytes(got[36:40]) and bytes(got[4:8]) == bytes(got[32:36])
                                                                   and bytes(got[8:16]) == bytes(got[40:48]))
                    sh["exchange_one_rank_real_rccl"] = out
                    proj = fused_us + out["gather_us"]
                    sh["projected_c4_8gpu_us"] = round(proj, 2)
                    sh["projected_c4_8gpu_note"] = ("shard pass (fused sum + argmax over 128 MiB, cold, back to back) + the exchange's latency floor "
                                                    "(one all-gather on a 1-rank communicator + combine kernel); the xGMI hop of a 16-byte record per peer is not in it")
                    result["roofline"].update({"reduce_exchange_one_rank_gather_us": out["gather_us"], "reduce_exchange_one_rank_two_collectives_us": out["all_reduce_us"],
                                               "reduce_shard_pass_plus_exchange_us": out["shard_pass_plus_exchange_us"],
                                               "projected_c4_8gpu_us": round(proj, 2), "c4_8gpu_projected_us": round(proj, 2),
                                               "projected_c4_8gpu_GBs_whole_job": round(n_total * 4 / proj / 1e3, 1)})
                outcome = run_with_watchdog(exchange_floor, 120.0)
                if outcome is not None:
                    sh["exchange_one_rank_real_rccl"] = {"error": outcome}
            if world > 1:
                def exchange():
                    # C4 end to end (cubecl_amd/sharded.py): local fused pass over this rank's slice, then the
                    # exchange step -- RCCL all-reduce of the f32 partial sums (ServerCommunication::all_reduce)
                    # and an all-gather of the (max value, index) records; every rank runs the same combine.
                    from cubecl_amd import sharded
                    ids = [DeviceId(0, i) for i in range(world)]
                    if job.native is None:      # (native: the job's communicator IS the one for this device set -- comm_init returns at once)
                        if rank == 0:
                            store.set("c4/unique_id", bytes(client.comm_unique_id()))
                        client.comm_init(ids, bytes(store.get("c4/unique_id")), rank=rank)
                    from cubecl_amd import ReduceOperation
                    start, count = sharded.shard_aligned_range(n_total, rank, world, 4)
                    assert count == n_local
                    rec = outs.offset_end_by(outs.size - 16)                          # {f32 max, f32 partial sum, u64 local index}
                    g_sum = outs.offset_start_by(32).offset_end_by(outs.size - 36)    # the job's results, on every device
                    g_val = outs.offset_start_by(36).offset_end_by(outs.size - 40)
                    g_idx = outs.offset_start_by(40).offset_end_by(outs.size - 48)
                    ex = sharded.RcclExchange(client, ids, rank)
                    starts = [sharded.shard_aligned_range(n_total, r, world, 4)[0] for r in range(world)]

                    def e2e(mode="gather"):
                        # local fused pass -> ONE all-gather of the 16-byte records on the communication stream -> comm -> compute
                        # fence -> 64-lane combine kernel (partial sums added in rank order, argmax candidates folded): sum,
                        # maximum and its global index are in device memory on every rank when the stream drains, all inside
                        # the timer.  ("all_reduce": the reference's shape, all_reduce(Sum) + all-gather, timed beside it.)
                        client._s.check(lib.mi355_sum_argmax_f32(ctx, None, p_in, n_local, p_sum, p_val, p_idx, p_ws, ws.size))
                        ex.exchange_on_device(rec, starts, g_sum, g_val, g_idx, mode=mode)
                    dt2 = job_seconds(lambda: e2e("all_reduce"), iters=20, warmup=3)
                    dt = job_seconds(e2e, iters=20, warmup=3)
                    import numpy as np
                    gsum = float(np.frombuffer(client.read_one(g_sum), dtype=np.float32)[0])
                    gval = float(np.frombuffer(client.read_one(g_val), dtype=np.float32)[0])
                    gidx = int(np.frombuffer(client.read_one(g_idx), dtype=np.uint64)[0])
                    # cross-check outside the timer: the host rule over the same gathered records (cubecl_amd/sharded.py)
                    raw = np.frombuffer(client.read_one(ex._buf.offset_start_by(64)), dtype=np.uint8)[: 16 * world].reshape(world, 16)
                    pairs = [(float(raw[r, 0:4].copy().view(np.float32)[0]), starts[r] + int(raw[r, 8:16].copy().view(np.uint64)[0]))
                             for r in range(world)]
                    hval, hidx = sharded.combine_argmax(pairs)
                    res["sharded_sum_argmax_exchange"] = {"ms": round(dt * 1e3, 4),
                                                          "GBs_total": round(n_total * 4 / dt / 1e9, 1),
                                                          "sum": gsum, "argmax_index": gidx, "argmax_value": gval,
                                                          "device_combine_equals_host_rule": bool(gidx == hidx and (gval == hval or (gval != gval and hval != hval))),
                                                          "exchange": "ONE RCCL all-gather (16 B per rank: max, partial sum, index) + combine kernel",
                                                          "ms_with_all_reduce_and_all_gather": round(dt2 * 1e3, 4)}
                    result["roofline"].update({"reduce_sum_argmax_exchange_ms": round(dt * 1e3, 4),
                                               "reduce_sum_argmax_exchange_GBs_whole_job": round(n_total * 4 / dt / 1e9, 1),
                                               "reduce_sum_argmax_exchange_two_collectives_ms": round(dt2 * 1e3, 4)})
                # The exchange cannot be rehearsed on the single-GPU pod: never let it take the headline line down with
                # it.  It runs under a watchdog; a rank that does not come back within the limit reports so and the
                # process leaves through os._exit after printing (a hung collective cannot be cancelled).
                outcome = run_with_watchdog(exchange, 180.0)
                if outcome is not None:
                    res["sharded_sum_argmax_exchange"] = {"error": outcome}
            return res

        def book_reduce():
            # the reference's only published reduce cases (cubecl-book getting-started: sum over the last axis;
            # 1.085 ms / 3.124 ms / 1.483 ms / 0.924 ms on an unnamed wgpu device)
            out = {}
            for shape, ref_ms in (((512, 8192), 1.085), ((128, 32768), 3.124), ((64, 256, 1024), 1.483), ((64, 64, 4096), 0.924)):
                n = 1
                for d_ in shape:
                    n *= d_
                cols = shape[-1]
                x = TensorHandle.uniform(client, (n,), ElemType.F32, SEED, 800, 0.0, 1.0)
                y = client.empty(n // cols * 4)
                med, best = samples_op(client, ev, lambda: client._s.check(lib.mi355_reduce_last_axis_sum_f32(
                    ctx, None, x.device_ptr(), y.device_ptr(), n // cols, cols, cols)))
                out["x".join(map(str, shape))] = {"median_us": round(med * 1e3, 2), "GBs": round(n * 4 / med / 1e6, 1),
                                                   "book_wgpu_ms": ref_ms}
            return out

        def reduce_axes():
            # reductions over one axis of a contiguous tensor and the operations beside sum / argmax -- HBM-bound, the input
            # read once; median of 15 samples (reference protocol), GB/s of input
            from cubecl_amd import ops
            out = {}
            for shape, axis in (((8192, 8192), 0), ((8192, 8192), 1), ((4, 65536, 1024), 1), ((16384, 16384), 0), ((64, 256, 1024), 1)):
                n = 1
                for d_ in shape:
                    n *= d_
                m = n // shape[axis]
                x = TensorHandle.uniform(client, shape, ElemType.F32, SEED, 820, -1.0, 1.0)
                o = TensorHandle.new_contiguous((m,), client.empty(m * 4), ElemType.F32)
                oi = TensorHandle.new_contiguous((m,), client.empty(m * 4), ElemType.U32)
                ent = {}
                for op in ("sum", "max", "argmax"):
                    fn = (lambda: ops.argreduce_axis(client, x, oi, axis, op)) if op == "argmax" else (lambda: ops.reduce_axis(client, x, o, axis, op))
                    med, _ = samples_op(client, ev, fn)
                    ent[op] = {"median_us": round(med * 1e3, 1), "GBs": round(n * 4 / med / 1e6, 1)}
                out["x".join(map(str, shape)) + f"_axis{axis}"] = ent
                del x, o, oi
            n = 1 << 28
            x = TensorHandle.uniform(client, (n,), ElemType.F32, SEED, 300, 0.0, 1.0)
            o = TensorHandle.new_contiguous((1,), client.empty(8), ElemType.F32)
            oi = TensorHandle.new_contiguous((1,), client.empty(8), ElemType.U64)
            ent = {}
            for op in ("max", "min", "mean", "prod", "argmin"):
                fn = (lambda: ops.argreduce(client, x, oi, None, op)) if op == "argmin" else (lambda: ops.reduce(client, x, o, op))
                b2b = time_op(client, ev, fn, 20, warmup=3)
                ent[op] = {"back_to_back_us": round(b2b * 1e3, 1), "GBs": round(n * 4 / b2b / 1e6, 1), "frac_of_8TBs": round(n * 4 / b2b / 1e6 / PEAK_HBM_GBS, 4)}
            out["array_wide_1GiB_f32"] = ent
            return out

        def sum_things_c1():
            # config C1: the reference's own CPU-runnable case (examples/sum_things: array-wide sum of 2^20 f32) -- launch-bound on a GPU
            n = 1 << 20
            x = TensorHandle.uniform(client, (n,), ElemType.F32, SEED, 810, 0.0, 1.0)
            ws = client.empty(1 << 17)
            o = client.empty(64)
            call = lambda: client._s.check(lib.mi355_reduce_sum_f32(ctx, None, x.device_ptr(), n, o.device_ptr(), ws.device_ptr(), ws.size))
            med, best = samples_op(client, ev, call)
            b2b = time_op(client, ev, call, 200)
            return {"elements": n, "median_us": round(med * 1e3, 2), "min_us": round(best * 1e3, 2), "back_to_back_us": round(b2b * 1e3, 2),
                    "GBs_back_to_back": round(n * 4 / b2b / 1e6, 1)}

        def gemm_f32_c2():
            M = 4096
            fa = TensorHandle.uniform(client, (M, M), ElemType.F32, SEED, 400, -1.0, 1.0)
            fb = TensorHandle.uniform(client, (M, M), ElemType.F32, SEED, 401, -1.0, 1.0)
            fc = client.empty(M * M * 4)
            out = {}
            for name, tb in (("NT", 1), ("NN", 0)):
                d = gemm_desc(N, M, M, M, N.DTYPE_F32, N.DTYPE_F32, trans_b=tb)
                med, best = samples_op(client, ev, lambda: client._s.check(
                    lib.mi355_gemm(ctx, None, C.byref(d), fa.device_ptr(), fb.device_ptr(), fc.device_ptr())))
                tf = 2.0 * M ** 3 / med / 1e9
                alg = C.c_int32()
                lib.mi355_gemm_select(ctx, C.byref(d), C.byref(alg))
                # the reference protocol syncs around every sample (the chip idles in between); also report
                # 20 launches back to back, the regime of the headline number
                b2b = time_op(client, ev, lambda: client._s.check(
                    lib.mi355_gemm(ctx, None, C.byref(d), fa.device_ptr(), fb.device_ptr(), fc.device_ptr())), 20)
                out[name] = {"median_ms": round(med, 4), "TFLOPs": round(tf, 1), "frac_of_157TF": round(tf / PEAK_F32_TFLOPS, 4),
                             "back_to_back_ms": round(b2b, 4), "back_to_back_TFLOPs": round(2.0 * M ** 3 / b2b / 1e9, 1),
                             "algo": alg.value}
                result["roofline"].update({f"c2_f32_4096_{name}_achieved_TFLOPs": round(2.0 * M ** 3 / b2b / 1e9, 1),
                                           f"c2_f32_4096_{name}_frac": round(2.0 * M ** 3 / b2b / 1e9 / PEAK_F32_TFLOPS, 4),
                                           f"c2_f32_4096_{name}_frac_per_sample_median": round(tf / PEAK_F32_TFLOPS, 4)})
            result["roofline"]["c2_f32_peak_TFLOPs"] = PEAK_F32_TFLOPS
            result["roofline"]["c2_f32_frac"] = result["roofline"]["c2_f32_4096_NT_frac"]
            return out

        def gemm_fp8():
            # not a BASELINE config: the fp8 variant of the headline shape, e4m3 in, bf16 out
            out = {}
            for S_ in (8192, 4096):
                qa = TensorHandle.uniform(client, (S_, S_), ElemType.F8E4M3, SEED, 900, -1.0, 1.0)
                qb = TensorHandle.uniform(client, (S_, S_), ElemType.F8E4M3, SEED, 901, -1.0, 1.0)
                qc = client.empty(S_ * S_ * 2)
                d = gemm_desc(N, S_, S_, S_, N.DTYPE_F8E4M3, N.DTYPE_BF16, trans_b=1)
                call = lambda: client._s.check(lib.mi355_gemm(ctx, None, C.byref(d), qa.device_ptr(), qb.device_ptr(), qc.device_ptr()))
                time_op(client, ev, call, 40)                                  # past the DVFS ramp
                b2b = time_op(client, ev, call, 30)
                tf = 2.0 * S_ ** 3 / b2b / 1e9
                out[f"{S_}^3"] = {"back_to_back_ms": round(b2b, 4), "TFLOPs": round(tf, 1), "frac_of_5PF": round(tf / 5000.0, 4)}
            return out

        def gemm_mx():
            # block-scaled variants: one ue8m0 scale per 32 k-values of each operand row, bf16 out;
            # the timed call includes the re-arrangement of the scales into the kernel's per-K-tile layout
            out = {}
            for name, dt, epb, peak in (("mxfp8_e4m3", ElemType.F8E4M3, 1, 5000.0), ("mxfp4_e2m1", ElemType.F4E2M1X2, 2, 10000.0)):
                S_ = 8192
                if epb == 1:
                    qa = TensorHandle.uniform(client, (S_, S_), dt, SEED, 910, -1.0, 1.0)
                    qb = TensorHandle.uniform(client, (S_, S_), dt, SEED, 911, -1.0, 1.0)
                else:                                              # uniformly random nibble pairs
                    qa = TensorHandle.uniform(client, (S_ * S_ // 2,), dt, SEED, 910, 0.0, 256.0)
                    qb = TensorHandle.uniform(client, (S_ * S_ // 2,), dt, SEED, 911, 0.0, 256.0)
                sa = TensorHandle.uniform(client, (S_ * S_ // 32,), ElemType.UE8M0, SEED, 912, 124.0, 131.0)
                sb = TensorHandle.uniform(client, (S_ * S_ // 32,), ElemType.UE8M0, SEED, 913, 124.0, 131.0)
                qc = client.empty(S_ * S_ * 2)
                d = N.GemmScaledDesc(m=S_, n=S_, k=S_, batch=1, lda=S_, ldb=S_, ldc=S_, ld_sa=S_ // 32, ld_sb=S_ // 32, dtype_a=int(dt),
                                     dtype_b=int(dt), dtype_c=N.DTYPE_BF16, block=32)
                call = lambda: client._s.check(lib.mi355_gemm_scaled(ctx, None, C.byref(d), qa.device_ptr(), sa.device_ptr(),
                                                                     qb.device_ptr(), sb.device_ptr(), qc.device_ptr()))
                time_op(client, ev, call, 40)
                b2b = time_op(client, ev, call, 30)
                tf = 2.0 * S_ ** 3 / b2b / 1e9
                out[name] = {"shape": f"{S_}^3", "back_to_back_ms": round(b2b, 4), "TFLOPs": round(tf, 1), "frac_of_dense_peak": round(tf / peak, 4)}
            return out

        def batched_c5():
            # config C5: batch 512 x 2048^3 bf16, the batch cut into contiguous runs by sharded.shard_range -- the SAME
            # 512-matrix job at every N (N = 1: all 512 on this GPU, 12 GiB of operands + results; N = 8: 64 each), so
            # "8 GPUs vs 1" compares one job with itself.  No data-path collective.
            from cubecl_amd import sharded
            total = 512
            M = 2048
            _, mine = sharded.shard_range(total, rank, world)
            ba = TensorHandle.uniform(client, (mine, M, M), ElemType.BF16, SEED, 500 + rank, -1.0, 1.0)
            bb = TensorHandle.uniform(client, (mine, M, M), ElemType.BF16, SEED, 600 + rank, -1.0, 1.0)
            bc = client.empty(mine * M * M * 2)

            def measure(count, tb, pb_handle):
                d = gemm_desc(N, M, M, M, N.DTYPE_BF16, N.DTYPE_BF16, trans_b=tb, batch=count)
                call = lambda: client._s.check(lib.mi355_gemm(ctx, None, C.byref(d), ba.device_ptr(), pb_handle.device_ptr(), bc.device_ptr()))
                med, best = samples_op(client, ev, call, samples=9, warmup=5)      # the first passes after another config ride the DVFS ramp
                # priced like the other extras on back-to-back passes (HIP events around 10 passes, best of 3): a per-sample event
                # pair with a sync on both sides lets the clock sag between samples; the per-sample median is kept beside it
                b2b = min(time_op(client, ev, call, 10, warmup=3) for _ in range(3))
                alg = C.c_int32()
                lib.mi355_gemm_select(ctx, C.byref(d), C.byref(alg))
                # whole-job figure (the >= 6x at 8 GPUs target is quoted on it): barrier -> every rank launches its run of
                # the batch 20 x back to back (after 5 untimed passes) -> sync -> the slowest rank's wall time
                job_s = job_seconds(call, iters=20, warmup=5)
                flop_rank = 2.0 * M ** 3 * count
                tf = flop_rank / b2b / 1e9
                return {"batch_this_rank": count, "algo": alg.value, "median_ms": round(med, 3), "back_to_back_ms": round(b2b, 3),
                        "TFLOPs_per_gpu": round(tf, 1), "TFLOPs_per_gpu_per_sample_median": round(flop_rank / med / 1e9, 1),
                        "frac_of_2.5PF": round(tf / PEAK_BF16_TFLOPS, 4), "job_ms_per_pass": round(job_s * 1e3, 3)}, job_s

            res, job_s = measure(mine, 1, bb)
            tf_job = 2.0 * M ** 3 * total / job_s / 1e12            # shard_range covers [0, 512) exactly once: all ranks' FLOP
            res.update({"batch_total": total, "sharding": f"sharded.shard_range({total}, rank, {world})",
                        "TFLOPs_total": round(tf_job, 1),
                        "frac_of_2.5PF_per_gpu_job": round(tf_job / world / PEAK_BF16_TFLOPS, 4)})
            c5_ent, c5_why = _pmc_entry("pmc_traffic.json", "gemm_bf16_c5_batch512", "gemm_q", C5_KERNEL) if (world == 1 and res["algo"] == C5_ALGO) else (None, "pass is for the 1-GPU job on lp256qm")
            result["roofline"].update({
                "c5_whole_job_TFLOPs": round(tf_job, 1), "c5_frac": round(tf_job / world / PEAK_BF16_TFLOPS, 4),
                "c5_batched_512x2048_achieved_TFLOPs_whole_job": round(tf_job, 1), "c5_batched_512x2048_frac": round(tf_job / world / PEAK_BF16_TFLOPS, 4),
                "c5_batched_512x2048_kernel_ms": res["back_to_back_ms"], "c5_batched_512x2048_frac_per_sample_median": round(res["TFLOPs_per_gpu_per_sample_median"] / PEAK_BF16_TFLOPS, 4),
                "c5_batched_512x2048_algorithmic_bytes": 3 * mine * M * M * 2,
                "c5_batched_512x2048_traffic": (c5_ent or {}).get("hbm_bytes_per_launch"), "c5_batched_512x2048_l2_hit_rate": (c5_ent or {}).get("l2_hit_rate"),
                "c5_batched_512x2048_mfma_util_pmc": (c5_ent or {}).get("mfma_util")})
            result["roofline"]["batched_512x2048"] = {"achieved": round(tf_job, 1), "frac": round(tf_job / world / PEAK_BF16_TFLOPS, 4),
                                                      "traffic": (c5_ent or {}).get("hbm_bytes_per_launch")}
            res["pmc"] = ({k: c5_ent.get(k) for k in ("hbm_bytes_per_launch", "fetch_bytes", "write_bytes", "l2_hit_rate", "mfma_util", "git_sha", "source_sha", "date")}
                          if c5_ent else c5_why)
            # the reference's default operand layout (TensorHandle::new_contiguous: rhs [K][N] row-major), same job
            nn, job_nn = measure(mine, 0, bb)
            nn["TFLOPs_total"] = round(2.0 * M ** 3 * total / job_nn / 1e12, 1)
            res["row_major_rhs_NN"] = nn
            if mine > 64:
                # the 64-matrix run one GPU of an 8-GPU job holds, on this GPU alone (the figure rounds 1-2 quoted)
                s64, _ = measure(64, 1, bb)
                res["shard_of_64"] = s64
            return res

        def skinny():
            out = {}
            shapes = [(8192, 8192, 64, 1), (64, 8192, 8192, 1), (8192, 64, 8192, 1), (1, 8192, 8192, 1), (16, 8192, 8192, 1), (16, 28672, 8192, 1),
                      (64, 28672, 8192, 1), (128, 28672, 8192, 1), (4096, 4096, 4096, 1), (6144, 6144, 6144, 1), (4608, 4096, 8192, 1),
                      (2048, 2048, 2048, 1), (4096, 2048, 4096, 1), (3072, 3072, 3072, 1),
                      # the reference's default rhs layout (row-major [K][N], TensorHandle::new_contiguous): staged natively, no re-layout
                      (8192, 8192, 8192, 0), (4096, 4096, 4096, 0), (2048, 2048, 2048, 0),
                      # ... and the decode case in that layout: few rows of x times a row-major weight [K][N]
                      (1, 8192, 8192, 0), (16, 8192, 8192, 0), (16, 28672, 8192, 0), (64, 28672, 8192, 0)]
            for (m, n, k, tb) in shapes:
                # Cold operands (advisor, round 2): a 128 MiB operand re-read by 20 back-to-back launches is partly served by the
                # 256 MiB Infinity Cache, which flatters HBM-bound shapes.  Launches rotate through as many operand sets as it
                # takes to exceed 768 MiB in total (at most 8), so every launch finds its operands in HBM; and the figure is the
                # MEDIAN of five back-to-back runs, not their minimum.
                fp = 2 * (m * k + n * k + m * n)
                nsets = max(1, min(8, -(-(768 << 20) // fp)))
                sets = []
                for i in range(nsets):
                    sets.append((TensorHandle.uniform(client, (m, k), ElemType.BF16, SEED, 700 + 2 * i, -1.0, 1.0),
                                 TensorHandle.uniform(client, (n, k), ElemType.BF16, SEED, 701 + 2 * i, -1.0, 1.0),    # ([K][N] when tb == 0: same bytes, other meaning)
                                 client.empty(m * n * 2)))
                d = gemm_desc(N, m, n, k, N.DTYPE_BF16, N.DTYPE_BF16, trans_b=tb)
                alg = C.c_int32()
                lib.mi355_gemm_select(ctx, C.byref(d), C.byref(alg))
                turn = [0]

                def call():
                    sa, sb, sc = sets[turn[0] % nsets]
                    turn[0] += 1
                    client._s.check(lib.mi355_gemm(ctx, None, C.byref(d), sa.device_ptr(), sb.device_ptr(), sc.device_ptr()))
                med, _ = samples_op(client, ev, call, samples=7, warmup=2)
                # these launches are tens of microseconds: a per-sample event pair adds one launch gap (~2.5 us) to each, so the
                # rate is priced on 20 back-to-back launches (like the roofline objects) and the per-sample median is kept beside it
                runs = sorted(time_op(client, ev, call, 20, warmup=3) for _ in range(5))
                b2b = runs[2]
                # ... and, for the shapes small enough to be cache-assisted, the figure on ONE operand set (what rounds 1-2 quoted)
                warm = None
                if nsets > 1:
                    one = lambda: client._s.check(lib.mi355_gemm(ctx, None, C.byref(d), sets[0][0].device_ptr(), sets[0][1].device_ptr(), sets[0][2].device_ptr()))
                    warm = sorted(time_op(client, ev, one, 20, warmup=3) for _ in range(3))[1]
                ra, rb = C.c_int32(), C.c_int32()
                lib.mi355_gemm_relayout_plan(C.byref(d), C.byref(ra), C.byref(rb))
                out[f"{m}x{n}x{k}" + ("" if tb else "_NN")] = {"median_ms": round(med, 4), "back_to_back_ms": round(b2b, 4), "TFLOPs": round(2.0 * m * n * k / b2b / 1e9, 1),
                                                                "algo": alg.value, "algorithmic_GBs": round(2.0 * (m * k + n * k + m * n) / b2b / 1e6, 1),
                                                                "operands_relaid_out": bool(ra.value or rb.value), "operand_sets_rotated": nsets,
                                                                "back_to_back_min_ms": round(runs[0], 4),
                                                                "warm_one_operand_set_ms": round(warm, 4) if warm else None}
            return out

        def contiguous():
            # into_contiguous (crates/cubecl-std/src/tensor/contiguous/launch.rs:5-20) of 512 MiB views: HBM-bound,
            # algorithmic bytes = read + write = 2 x elements x elem_size; roofline = the 8 TB/s HBM peak, and the
            # measured copy ceiling of this board is reported next to it under measured_ceilings.
            from cubecl_amd import ops
            names = ["flat", "rows", "transpose", "generic", "two_sided"]
            out = {}
            cases = (("bf16_transpose_16384x16384", ElemType.BF16, [16384, 16384], [1, 16384]),
                     ("bf16_transpose_16640x15872", ElemType.BF16, [15872, 16640], [1, 15872]),
                     ("f32_transpose_8192x16384", ElemType.F32, [16384, 8192], [1, 16384]),
                     ("bf16_batched_transpose_64x2048x2048", ElemType.BF16, [64, 2048, 2048], [2048 * 2048, 1, 2048]),
                     ("bf16_kT_1024x4096x64", ElemType.BF16, [1024, 64, 4096], [4096 * 64, 1, 64]),
                     ("f32_nchw_to_nhwc_167x256x56x56", ElemType.F32, [167, 56, 56, 256], [256 * 3136, 56, 1, 3136]),
                     ("f32_pitched_rows_32768x4096_of_4160", ElemType.F32, [32768, 4096], [4160, 1]),
                     ("u8_nhwc_to_nchw_3_channels", ElemType.U8, [3566, 3, 224, 224], [3 * 224 * 224, 1, 224 * 3, 3]),
                     ("f32_contiguous_512MiB", ElemType.F32, [1 << 27], [1]))
            for name, dt, shape, strides in cases:
                n = 1
                for d_ in shape:
                    n *= d_
                span = sum((d_ - 1) * s_ for d_, s_ in zip(shape, strides)) + 1
                src = client.empty(span * dt.size())
                dst = client.empty(n * dt.size())
                tin = TensorHandle.new(src, shape, strides, dt)
                tout = TensorHandle.new_contiguous(shape, dst, dt)
                path, access = ops.copy_plan(client, tin, tout)
                med, best = samples_op(client, ev, lambda: ops.copy_into(client, tin, tout), samples=9, warmup=3)
                moved = 2 * n * dt.size()
                out[name] = {"mover": names[path], "access_bytes": access, "median_us": round(med * 1e3, 1),
                             "GBs": round(moved / med / 1e6, 1), "frac_of_8TBs": round(moved / med / 1e6 / PEAK_HBM_GBS, 4)}
            return out


        # Order of the line: the sections outside BASELINE.json first, its configs C1, C2, C5, C4 last -- the driver keeps the
        # LAST 8 KB of stdout (review of round 3, weak #7) -- and the compact per-config summary at the very end.
        guarded("into_contiguous_512MiB", contiguous)
        guarded("gemm_bf16_shapes", skinny)
        guarded("gemm_fp8_e4m3", gemm_fp8)
        guarded("gemm_block_scaled", gemm_mx)
        guarded("book_reduce_last_axis_f32", book_reduce)
        guarded("reduce_axes_and_ops_f32", reduce_axes)
        guarded("measured_ceilings", probes)
        guarded("headline_kernel_operand_sensitivity", operand_sensitivity)
        guarded("sum_things_1M_f32", sum_things_c1)
        guarded("gemm_f32_4096", gemm_f32_c2)
        guarded("batched_gemm_2048_bf16", batched_c5)
        guarded("reduce_1GiB_f32", reduce_c4)

    if extra:
        result["extra"] = extra
    if errors:
        result["extra_errors"] = errors
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        def farewell():
            barrier()
            if job.dist is not None:
                dist.destroy_process_group()
        if HUNG or run_with_watchdog(farewell, 60.0) is not None or HUNG:
            sys.stdout.flush()
            os._exit(0)          # a collective is stuck somewhere: the line is out, leave without waiting for it


if __name__ == "__main__":
    main()
