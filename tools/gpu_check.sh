#!/bin/bash
# Runs on the GPU box (via gpurun): GPU tests, smoke, bench, rocprof summary. Outputs -> gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
TAG=${1:-run}
echo "== rocm-smi"; rocm-smi --showproductname 2>/dev/null | head -8
echo "== pytest -m gpu"
# (SKIP_PYTEST=1: the suite already ran on this tree in an earlier call of the same evidence round -- e.g. the one that took the PMC passes,
#  whose json files must sit in profiles/ BEFORE the bench line is taken; its log is kept)
if [ "${SKIP_PYTEST:-0}" != 1 ]; then
timeout 1200 python -m pytest tests -m gpu -q --no-header --timeout 300 -p no:cacheprovider --maxfail=60 > $OUT/${TAG}_pytest.log 2>&1
echo "pytest exit $?"; tail -n 40 $OUT/${TAG}_pytest.log
fi
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${TAG}_smoke.log 2>&1; echo "smoke exit $?"; tail -n 5 $OUT/${TAG}_smoke.log
echo "== bench"
timeout 900 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench exit $?"; cat $OUT/${TAG}_bench.json; tail -n 5 $OUT/${TAG}_bench.err
echo "== rocprof"
export TMPDIR=/tmp
# headline only, the default step counts: the GEMM kernel's average duration in the stats must agree with roofline.kernel_ms
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/${TAG}_prof" -o bench -- python "$OLDPWD/bench.py" --no-extras --no-cpu-baseline > "$OLDPWD/$OUT/${TAG}_prof.log" 2>&1 ); echo "rocprof exit $?"
# everything (extras included), fewer steps: per-kernel time of the other configs (SKIP_ALL=1 leaves this pass out)
[ "${SKIP_ALL:-0}" = 1 ] || ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/${TAG}_prof_all" -o bench -- python "$OLDPWD/bench.py" --steps 10 --warmup 3 --no-cpu-baseline > "$OLDPWD/$OUT/${TAG}_prof_all.log" 2>&1 ); echo "rocprof(all) exit $?"
# the 1 GiB reductions on the same binary (the reduce half of the metric): 20 launches of each kernel
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/${TAG}_prof_reduce" -o reduce -- python "$OLDPWD/tools/reduce_probe.py" > "$OLDPWD/$OUT/${TAG}_prof_reduce.log" 2>&1 ); echo "rocprof(reduce) exit $?"
find $OUT/${TAG}_prof -name "*kernel_stats*" | head -3
f=$(find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -n 25 "$f"
