#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
for rep in 1 2; do for so in libmi355cube.so variants/libmi355cube_pol1.so variants/libmi355cube_pol2.so variants/libmi355cube_pol3.so variants/libmi355cube_pol4.so; do
  echo "== $so"; MI355CUBE_LIB=$PWD/cubecl_amd/csrc/$so timeout 300 python tools/ab_algos.py --rounds 5 --algos lp128,lp256x128,stream64 2048x2048x2048 4096x2048x4096 2048x2048x8192 64x8192x8192 16x8192x8192 2>&1 | tail -5
done; done 2>&1 | tee $OUT/r03m_policy.txt
