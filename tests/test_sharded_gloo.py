"""The N > 1 path without GPUs: the sharding + combine logic of cubecl_amd/sharded.py run by two
(and three) real processes over torch.distributed / gloo.  The local pass is played by the CPU
oracle here (test infrastructure); on GPUs it is one mi355_sum_argmax_f32 launch per rank and the
exchange runs over RCCL -- the partition and combine code under test is the same.

Mirrors the reference's multi-device tests, which compare against a closed form of the per-device
contributions (crates/cubecl-core/src/runtime_tests/all_reduce.rs:5-62).
"""
import os
import socket
import struct
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from cubecl_amd import sharded  # noqa: E402


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make_case(case: str) -> np.ndarray:
    import oracle
    if case == "uniform":
        return oracle.fill_uniform(100_003, 31, 0.0, 1.0)
    if case == "tie_across_shards":          # the maximum appears in both halves: the lower index must win
        x = oracle.fill_uniform(65_536, 32, -1.0, 0.5)
        x[60_000] = 7.0
        x[1_234] = 7.0
        return x
    if case == "nan_in_last_shard":          # NaN ranks above every number; first NaN wins
        x = oracle.fill_uniform(40_000, 33, -1.0, 1.0)
        x[39_990] = np.float32("nan")
        x[39_995] = np.float32("nan")
        x[5] = np.float32("inf")
        return x
    if case == "signed_zero":                # -0 == +0: equal keys keep the lower index
        x = np.full(10_000, -3.0, dtype=np.float32)
        x[9_000] = np.float32(0.0)
        x[2_000] = np.float32(-0.0)
        return x
    if case == "tiny":                       # fewer aligned blocks than ranks: some shards are empty
        return np.array([1.0, 5.0, -2.0], dtype=np.float32)
    raise ValueError(case)


CASES = ["uniform", "tie_across_shards", "nan_in_last_shard", "signed_zero", "tiny"]


def _worker(rank: int, world: int, port: int, case: str, out_dir: str) -> None:
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        x = _make_case(case)

        def local_pass(start, count):
            part = np.ascontiguousarray(x[start:start + count])
            idx, val = oracle.argmax(part)
            return float(np.float32(oracle.sum_f64(part))), float(val), int(idx)

        ex = sharded.TorchExchange("cpu")
        res = sharded.sharded_sum_argmax(x.size, ex, local_pass)
        b0, bc = sharded.sharded_batch(13, ex)
        ex.barrier()
        bits = struct.unpack("<I", struct.pack("<f", np.float32(res.max_value)))[0]
        Path(out_dir, f"r{rank}.txt").write_text(f"{res.total!r} {bits} {res.max_index} {b0} {bc}\n")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,case", [(2, c) for c in CASES] + [(3, "uniform"), (3, "tiny")])
def test_sharded_sum_argmax_over_gloo(tmp_path, oracle, world, case):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, case, str(tmp_path)), nprocs=world, join=True)
    x = _make_case(case)
    exp_idx, exp_val = oracle.argmax(x)
    exact = oracle.sum_f64(x[np.isfinite(x)]) if case == "nan_in_last_shard" else oracle.sum_f64(x)
    covered = []
    for r in range(world):
        total, bits, idx, b0, bc = Path(tmp_path, f"r{r}.txt").read_text().split()
        # every rank ends with the same answer
        assert int(idx) == exp_idx, f"rank {r}: argmax index {idx} != {exp_idx}"
        got = struct.unpack("<f", struct.pack("<I", int(bits)))[0]
        assert (np.isnan(got) and np.isnan(exp_val)) or np.float32(got) == np.float32(exp_val)
        if case != "nan_in_last_shard":
            scale = max(float(oracle.sum_abs_f64(x)), 1e-30)
            assert abs(float(total) - exact) <= 1e-5 * scale     # BASELINE.json: 1e-5 relative for f32
        else:
            assert np.isnan(float(total)) or np.isinf(float(total))
        covered += list(range(int(b0), int(b0) + int(bc)))
    assert covered == list(range(13))    # batch shards tile [0, 13) exactly once, in rank order


def test_shard_ranges_cover_exactly_once():
    for n in (0, 1, 7, 8, 1 << 28, (1 << 28) + 3):
        for world in (1, 2, 3, 8):
            spans = [sharded.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            for (s0, c0), (s1, _) in zip(spans, spans[1:]):
                assert s0 + c0 == s1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
            al = [sharded.shard_aligned_range(n, r, world, 4) for r in range(world)]
            assert sum(c for _, c in al) == n
            for (s0, c0), (s1, _) in zip(al, al[1:]):
                assert s0 + c0 == s1 and s1 % 4 == 0 or s1 == n
    with pytest.raises(ValueError):
        sharded.shard_range(4, 2, 2)


def test_combine_argmax_matches_device_rule(oracle):
    # the host combine uses the same key as the oracle / the HIP kernel
    vals = [0.0, -0.0, 1.5, -1.5, float("inf"), float("-inf"), float("nan"), 3.4e38, -3.4e38, 1e-45]
    for v in vals:
        assert sharded.argmax_key(v) == oracle.lib().oracle_argmax_key(v)
    assert sharded.combine_argmax([(1.0, 10), (1.0, 3), (0.5, 0)]) == (1.0, 3)
    assert sharded.combine_argmax([(-0.0, 4), (0.0, 9)]) == (-0.0, 4)
    v, i = sharded.combine_argmax([(float("inf"), 1), (float("nan"), 8), (float("nan"), 5)])
    assert np.isnan(v) and i == 5
    assert sharded.combine_argmax([(2.0, -1), (1.0, -1)]) == (float("-inf"), 0)      # all shards empty
    assert sharded.combine_argmax([(2.0, -1), (1.0, 7)]) == (1.0, 7)


def test_local_exchange_is_identity():
    ex = sharded.LocalExchange()
    res = sharded.sharded_sum_argmax(10, ex, lambda s, c: (4.5, 2.0, 3))
    assert (res.total, res.max_value, res.max_index) == (4.5, 2.0, 3)
