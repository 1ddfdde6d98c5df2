#!/bin/bash
# usage (GPU box): tools/dev/ab_gemm.sh "<algo list>" [size] -- interleaved A/B of the headline GEMM across algos
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ALGOS=${1:-"4 5"}; SIZE=${2:-8192}
for rep in 1 2 3; do
for a in $ALGOS; do
  v=$(python bench.py --no-extras --no-cpu-baseline --size $SIZE --steps 20 --warmup 5 --algo $a 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['config']['kernel'])" 2>&1 | tail -1)
  echo "algo=$a size=$SIZE -> $v"
done
done
