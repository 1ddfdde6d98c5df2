#!/bin/bash
# usage (GPU box): tools/dev/sustain_variants.sh [algo] -- steady-state TF (median of blocks 10..59) per library variant
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ALGO=${1:-0}
for rep in 1 2; do
for so in cubecl_amd/csrc/libmi355cube.so cubecl_amd/csrc/variants/*.so; do
  MI355CUBE_LIB=$PWD/$so python tools/dev/sustain.py $ALGO | python -c "
import sys
v=[float(x) for x in sys.stdin.read().split(':')[1].split()]
s=sorted(v[10:]); print('$(basename $so)', 'first block', v[0], 'steady median', s[len(s)//2], 'min', s[0], 'max', s[-1])"
done
done
