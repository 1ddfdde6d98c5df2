#!/bin/bash
# usage (GPU box): tools/dev/q_variants.sh "spec spec ..." -- q_ab.py for the product .so and every variant .so
cd "${GRAFT_REPO_ROOT:-/root/repo}"
SPECS=${1:-2048x2048x2048x64}
for rep in 1 2; do
for so in cubecl_amd/csrc/libmi355cube.so cubecl_amd/csrc/variants/*.so; do
  echo "== $(basename $so)"; MI355CUBE_LIB=$PWD/$so timeout 300 python tools/dev/q_ab.py $SPECS 2>&1 | tail -n 4
done; done
