"""bench.py's launcher contract, checked without a GPU (the measurement itself needs the MI355X: tests/test_gpu_*.py and the
driver).  The review of round 2 found that `python bench.py --gpus 8` silently ran ONE rank and printed "n_gpus": 1; the plain
form now becomes its own launcher, and a line can never claim another rank count than the job had."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def _run(args, env_extra=None, timeout=600):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, capture_output=True, text=True, env=env, timeout=timeout)


def test_plain_multi_gpu_form_spawns_one_rank_per_gpu_and_never_prints_a_one_rank_line():
    """No GPU here, so every rank stops at "needs a GPU" -- but there must be TWO of them (self_spawn -> torch.distributed.run),
    the exit code must say failure, and no JSON line may appear (least of all one with n_gpus 1)."""
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-extras", "--no-cpu-baseline"])
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    # two ranks: both said so -- or one did and the launcher's failure report names the other (it SIGTERMs the slower rank
    # as soon as the first one exits: on a loaded box the second message may never be printed)
    said = r.stderr.count("bench.py needs a GPU")
    assert said >= 2 or (said == 1 and "local_rank: 0" in r.stderr and "local_rank: 1" in r.stderr), r.stderr[-2000:]


def test_rank_count_of_the_launcher_wins_over_a_contradicting_flag():
    r = _run(["--gpus", "4", "--no-extras"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_committed_two_rank_rehearsal_line_is_a_two_rank_job_over_the_whole_batch():
    """profiles/r03_rehearse_n2.json: `python bench.py --gpus 2` on the 1-GPU box (two ranks on one device, gloo for torch's own
    collectives -- tools/gpu_rehearse_n2.sh).  What the driver's scaling run will rely on: n_gpus = 2, C5 is the 512-matrix job
    cut by shard_range, job-level rates present."""
    f = ROOT / "profiles" / "r03_rehearse_n2.json"
    if not f.exists():
        pytest.skip("no rehearsal line committed yet")
    line = [l for l in f.read_text().splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["n_gpus"] == 2 and r["scaling"] == "weak"
    c5 = r["extra"]["batched_gemm_2048_bf16"]
    assert c5["batch_total"] == 512 and c5["batch_this_rank"] == 256 and c5["TFLOPs_total"] > 0
    assert r["extra"]["reduce_1GiB_f32"]["sum"]["GBs_total"] > 0


def test_guard_check_case_list_builds_without_a_device():
    """tools/guard_check.py --list: the descriptors the guard-page run walks (the random draws of the fuzz tests, the few-row / decode shapes
    in both rhs layouts, the benchmark's shapes) can be enumerated on any box -- the tool must not rot between GPU runs."""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    out = subprocess.run([sys.executable, str(root / "tools" / "guard_check.py"), "--list"], capture_output=True, text=True, timeout=120)
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert out.returncode == 0 and len(lines) >= 219, (out.stdout[-300:], out.stderr[-500:])
    assert any("8192, 8192, 8192" in l for l in lines) and any("28672" in l for l in lines)


# ---- bench.py --gpus 2 to completion without a device (review of round 3, next #7b) ---------------------------------------------------
FAKE = ROOT / "tests" / "fake_hip"
CSRC = ROOT / "cubecl_amd" / "csrc"


def _build_bench_libs():
    """(library, fake RCCL): the product's host runtime (runtime.cpp, pool.cpp, comm.cpp) on the fake HIP runtime, host stand-ins for
    the compute entry points bench.py's N > 1 control flow touches (tests/fake_hip/fake_kernels.cpp) and a generated WEAK stub
    returning MI355_E_UNSUPPORTED for every other prototype of the header (cubecl_amd._native refuses a library that lacks one); the
    multi-process stand-in for librccl.so.1 (tests/fake_hip/rccl_mp/)."""
    import re
    so, rccl = FAKE / "libbenchtest.so", FAKE / "rccl_mp" / "librccl.so.1"
    hdr = (ROOT / "include" / "mi355cube.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    stubs = ['#include "../../cubecl_amd/csrc/internal.hpp"', 'extern "C" {']
    for ret, name, params in re.findall(r"^([A-Za-z_][\w \t\*]*?)\b(mi355_\w+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.M):
        ret = ret.strip()
        body = "return 0;" if "*" in ret else "return MI355_E_UNSUPPORTED;"
        stubs.append(f'__attribute__((weak, visibility("default"))) {ret} {name}({" ".join(params.split())}) {{ {body} }}')
    stubs.append("}")
    gen = FAKE / "_gen_stubs.cpp"
    gen_text = "\n".join(stubs) + "\n"
    if not gen.exists() or gen.read_text() != gen_text:
        gen.write_text(gen_text)
    srcs = [CSRC / "runtime.cpp", CSRC / "pool.cpp", CSRC / "comm.cpp", FAKE / "fake_hip.cpp", FAKE / "fake_kernels.cpp", gen]
    deps = srcs + [CSRC / "internal.hpp", FAKE / "hip" / "hip_runtime.h", ROOT / "include" / "mi355cube.h"]
    if not so.exists() or so.stat().st_mtime < max(d.stat().st_mtime for d in deps):
        subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wno-unused-parameter", "-Wno-format-truncation", "-DFAKE_WITH_RUNTIME", "-shared", "-fPIC",
                        "-Wl,-Bsymbolic", "-I", str(FAKE), "-o", str(so)] + [str(x) for x in srcs] + ["-ldl", "-lpthread"], check=True)
    src = FAKE / "rccl_mp" / "fake_rccl_mp.cpp"
    if not rccl.exists() or rccl.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-shared", "-fPIC", "-o", str(rccl), str(src), "-lpthread"], check=True)
    return so, rccl


def test_two_rank_job_runs_to_completion_on_the_native_collectives_without_a_device():
    """`python bench.py --gpus 2` end to end on this box: bench.py becomes its own launcher (torch.distributed.run), the two ranks
    find the launcher's TCP store, rank 0's unique id travels through it, both join the LIBRARY's communicator (comm.cpp over the
    multi-process RCCL stand-in), barrier and max-over-ranks run on it (no torch process group: `job_collectives` == "native"), the
    C4 extra runs local pass -> ONE all-gather of {max, partial sum, index} records -> combine kernel (RcclExchange.exchange_on_device;
    the reference's all-reduce + all-gather shape is timed beside it), and rank 0 prints ONE line with
    n_gpus == 2.  The figures of such a run mean nothing; its control flow is what the first real 8-GPU run will execute."""
    so, rccl = _build_bench_libs()
    n = 1 << 20
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-plateau-warmup", "--extras", "reduce_1GiB_f32",
              "--reduce-elements", str(n)],
             {"BENCH_NO_TORCH_CUDA": "1", "MI355CUBE_LIB": str(so), "MI355_RCCL_LIBRARY": str(rccl), "FAKE_HIP_DEVICES": "2", "OMP_NUM_THREADS": "1"},
             timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    assert line["config"]["job_collectives"] == "native", line["config"]
    assert "librccl" in line["config"] and str(rccl) in line["config"]["librccl"]
    assert not line.get("extra_errors"), line.get("extra_errors")
    red = line["extra"]["reduce_1GiB_f32"]
    assert red["elements_per_gpu"] == n // 2 and red["sum"]["GBs_total"] > 0
    ex = red["sharded_sum_argmax_exchange"]
    assert "error" not in ex, ex
    assert ex["device_combine_equals_host_rule"] is True
    assert ex["exchange"].startswith("ONE RCCL all-gather") and ex["ms_with_all_reduce_and_all_gather"] > 0
    # the stand-in fill is lo + (hi - lo) * ((i * 2654435761 + tensor * 97) % 1000) / 1000 with tensor = 300 + rank over each
    # rank's own slice: the all-reduced sum and the combined argmax are what two hosts computing alone would get
    import numpy as np
    parts, best = [], None
    for rank in range(2):
        i = np.arange(n // 2, dtype=np.uint64)
        x = (((i * np.uint64(2654435761) + np.uint64((300 + rank) * 97)) % np.uint64(1000)).astype(np.float32) / np.float32(1000.0)).astype(np.float32)
        parts.append(np.float32(x.astype(np.float64).sum()))
        j = int(np.argmax(x))
        cand = (float(x[j]), rank * (n // 2) + j)
        if best is None or cand[0] > best[0]:
            best = cand
    assert abs(ex["sum"] - float(parts[0] + parts[1])) <= 1e-3 * abs(float(parts[0] + parts[1]))
    assert ex["argmax_index"] == best[1] and abs(ex["argmax_value"] - best[0]) < 1e-6
    assert line["roofline"]["reduce_sum_argmax_exchange_ms"] > 0


def test_one_gpu_line_is_torch_free_and_the_record_keeps_the_whole_metric():
    """Review of round 5, weak #1 / #4 / #8 and next #4, on the fake runtime: the N = 1 path imports no torch (so the process maps ONE HIP
    runtime, the /opt/rocm one the library is built and tested against, and `config.libamdhip64` says which); the driver's record of the
    line keeps scalars of `config` and the first ~20 keys of `roofline` cut at 40 characters -- so the tolerance reading and the reduce
    half of the metric (C4: 1 GiB sum, the 128 MiB shard, the projected 8-GPU job), C5 and C2 must be flat scalars among the first keys."""
    so, rccl = _build_bench_libs()
    n = 1 << 20
    r = _run(["--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-plateau-warmup", "--extras", "reduce_1GiB_f32",
              "--reduce-elements", str(n)],
             {"BENCH_NO_TORCH_CUDA": "1", "MI355CUBE_LIB": str(so), "MI355_RCCL_LIBRARY": str(rccl), "FAKE_HIP_DEVICES": "1", "OMP_NUM_THREADS": "1"},
             timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and not line.get("extra_errors"), line.get("extra_errors")
    assert line["config"]["torch_in_process"] is False
    assert all(not isinstance(v, (dict, list)) for v in line["config"].values()), [k for k, v in line["config"].items() if isinstance(v, (dict, list))]
    keys = list(line["roofline"])
    assert keys[:6] == ["bound", "achieved", "peak", "unit", "frac", "traffic"]
    want = ["reduce_sum_achieved_GBs", "reduce_sum_frac", "reduce_sum_traffic", "reduce_sum_kernel_ms", "reduce_shard8_sum_us", "reduce_shard8_sum_frac",
            "reduce_shard8_fused_us", "reduce_shard8_fused_frac", "c4_8gpu_projected_us", "c5_whole_job_TFLOPs", "c5_frac", "c2_f32_frac"]
    assert all(k in keys[:20] for k in want), keys[:24]
    assert all(len(k) <= 40 for k in keys[:22]), [k for k in keys[:22] if len(k) > 40]
    rf = line["roofline"]
    assert rf["reduce_sum_achieved_GBs"] > 0 and rf["reduce_sum_frac"] > 0 and rf["reduce_shard8_fused_us"] > 0 and rf["reduce_shard8_sum_frac"] > 0
    assert rf["c4_8gpu_projected_us"] > 0                       # shard pass + the exchange on a one-rank communicator (the RCCL stand-in here)
    assert rf["c5_frac"] is None and rf["c2_f32_frac"] is None  # those extras were not asked for: present, in place, empty


def test_threaded_single_process_model_runs_to_completion_without_a_device():
    """`python bench.py --threads 2`: the reference's own process model (one process, one host thread + one context per device:
    crates/cubecl-common/src/device/handle/channel.rs:24-37; one communicator joined from the device threads: crates/cubecl-cuda/
    src/compute/server.rs:669-703) end to end on the fake runtime + the in-process RCCL stand-in: both device threads step the
    headline, meet at the host barrier, run config C4's local pass + one-collective exchange, and ONE line comes out with n_gpus 2 --
    so the first multi-GPU box can run either model (review of round 4, next #8)."""
    so, _ = _build_bench_libs()
    rccl = FAKE / "rccl" / "librccl.so.1"
    src = FAKE / "rccl" / "fake_rccl.cpp"
    if not rccl.exists() or rccl.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-shared", "-fPIC", "-fvisibility=hidden", "-pthread", "-o", str(rccl), str(src)], check=True)
    n = 1 << 20
    r = _run(["--threads", "2", "--steps", "2", "--warmup", "1", "--reduce-elements", str(n)],
             {"BENCH_NO_TORCH_CUDA": "1", "MI355CUBE_LIB": str(so), "MI355_RCCL_LIBRARY": str(rccl), "FAKE_HIP_DEVICES": "2", "OMP_NUM_THREADS": "1",
              "FAKE_RCCL_BLOCKING": "1", "FAKE_RCCL_BLOCKING_INIT": "1"},
             timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    assert line["config"]["process_model"].startswith("one process, one host thread")
    ex = line["extra"]["reduce_1GiB_f32"]["sharded_sum_argmax_exchange"]
    assert ex["every_device_holds_the_same_result"] is True and ex["ms"] > 0
    import numpy as np
    parts, best = [], None                      # the stand-in fill of tests/fake_hip/fake_kernels.cpp, tensor = 300 + device, as in the two-rank test
    for rank in range(2):
        i = np.arange(n // 2, dtype=np.uint64)
        x = (((i * np.uint64(2654435761) + np.uint64((300 + rank) * 97)) % np.uint64(1000)).astype(np.float32) / np.float32(1000.0)).astype(np.float32)
        parts.append(np.float32(x.astype(np.float64).sum()))
        j = int(np.argmax(x))
        cand = (float(x[j]), rank * (n // 2) + j)
        if best is None or cand[0] > best[0]:
            best = cand
    assert abs(ex["sum"] - float(parts[0] + parts[1])) <= 1e-3 * abs(float(parts[0] + parts[1]))
    assert ex["argmax_index"] == best[1] and abs(ex["argmax_value"] - best[0]) < 1e-6


def test_native_rendezvous_that_fails_on_every_rank_falls_back_to_torch_together():
    """No RCCL at all (MI355_RCCL_LIBRARY points nowhere and the soname is not on this box's path): comm_init fails on both ranks,
    they agree through the store and fall back to torch.distributed (gloo here) TOGETHER -- the line says so, the job completes."""
    so, _ = _build_bench_libs()
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-plateau-warmup", "--no-extras"],
             {"BENCH_NO_TORCH_CUDA": "1", "MI355CUBE_LIB": str(so), "FAKE_HIP_DEVICES": "2", "BENCH_DIST_BACKEND": "gloo", "OMP_NUM_THREADS": "1",
              "MI355_RCCL_LIBRARY": "/nonexistent/librccl.so.1", "BENCH_TEST_NO_SYSTEM_RCCL": "1"}, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["job_collectives"].startswith("torch:gloo (native refused"), line["config"]
