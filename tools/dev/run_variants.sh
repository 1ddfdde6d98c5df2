#!/bin/bash
# usage (GPU box): tools/dev/run_variants.sh [size] [algo] -- headline GEMM for the product .so and every variant .so
cd "${GRAFT_REPO_ROOT:-/root/repo}"
SIZE=${1:-8192}; ALGO=${2:-0}
for rep in 1 2; do
for so in cubecl_amd/csrc/libmi355cube.so cubecl_amd/csrc/variants/*.so; do
  v=$(MI355CUBE_LIB=$PWD/$so python bench.py --no-extras --no-cpu-baseline --size $SIZE --steps 30 --warmup 5 --algo $ALGO 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['config']['kernel'], 'clock', r['shader_clock_GHz'], 'frac@clock', r['frac_of_peak_at_clock'])")
  echo "$(basename $so) size=$SIZE -> $v"
done
done
