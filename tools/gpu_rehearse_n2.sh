#!/bin/bash
# Runs on the 1-GPU box: the N > 1 control flow of bench.py with two ranks on ONE device, torch collectives over gloo
# (BENCH_FORCE_DEVICE / BENCH_DIST_BACKEND are rehearsal hooks, never set by the driver).  The RCCL exchange of the C4 extra
# cannot work with two ranks on one device and must fail cleanly; the job-level C4 / C5 figures must be there.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
# The plain form: bench.py finds no launcher around it (WORLD_SIZE unset) and becomes one (self_spawn -> torch.distributed.run).
BENCH_FORCE_DEVICE=0 BENCH_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 10 --warmup 3 > $OUT/rehearse_n2.json 2> $OUT/rehearse_n2.err
echo "rehearsal exit $?"
python - <<'PY'
import json
line = [l for l in open("gpurun_out/rehearse_n2.json") if l.startswith("{")][-1]
r = json.loads(line)
ex = r.get("extra", {})
print(json.dumps({"value": r["value"], "n_gpus": r["n_gpus"], "ms_per_step": r["ms_per_step"],
                  "c5": ex.get("batched_gemm_2048_bf16"), "c4_sum": ex.get("reduce_1GiB_f32", {}).get("sum"),
                  "c4_exchange": ex.get("reduce_1GiB_f32", {}).get("sharded_sum_argmax_exchange"), "errors": r.get("extra_errors")}, indent=1))
PY
tail -n 5 $OUT/rehearse_n2.err
