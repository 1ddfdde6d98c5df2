#!/bin/bash
# round 3, second GPU call: the row-major-B (NN) tests, then the bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_full_size.py -m gpu -q --no-header --timeout 300 -p no:cacheprovider --maxfail=40 -k "row_major or native or relayout or select" > $OUT/r03b_pytest_nn.log 2>&1
echo "pytest(nn) exit $?"; tail -n 25 $OUT/r03b_pytest_nn.log
timeout 900 python bench.py --no-cpu-baseline > $OUT/r03b_bench.json 2> $OUT/r03b_bench.err; echo "bench exit $?"; tail -n 3 $OUT/r03b_bench.err
python - <<'PY'
import json
r=json.loads([l for l in open('gpurun_out/r03b_bench.json') if l.startswith('{')][-1])
ex=r['extra']
print('value',r['value'])
print(json.dumps(ex['batched_gemm_2048_bf16'],indent=0)[:1800])
for k,v in ex['gemm_bf16_shapes'].items(): print(k,v)
print(r.get('extra_errors'))
PY
