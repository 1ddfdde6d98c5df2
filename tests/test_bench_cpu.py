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
    assert r.stderr.count("bench.py needs a GPU") >= 2, r.stderr[-2000:]


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
