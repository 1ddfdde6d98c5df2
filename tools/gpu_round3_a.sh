#!/bin/bash
# round 3, first GPU call: the full GPU suite, the bench line, the two-rank rehearsal of the plain `--gpus 2` form
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
rm -f $OUT/parity_margins.jsonl
timeout 1500 python -m pytest tests -m gpu -q --no-header --timeout 300 -p no:cacheprovider --maxfail=30 -s > $OUT/r03a_pytest.log 2>&1
echo "pytest exit $?"; grep -v PARITY_MARGIN $OUT/r03a_pytest.log | tail -n 15
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/r03a_smoke.log 2>&1; echo "smoke exit $?"; tail -n 6 $OUT/r03a_smoke.log
timeout 900 python bench.py > $OUT/r03a_bench.json 2> $OUT/r03a_bench.err; echo "bench exit $?"; tail -c 600 $OUT/r03a_bench.json; tail -n 5 $OUT/r03a_bench.err
bash tools/gpu_rehearse_n2.sh 2>&1 | tail -n 60
