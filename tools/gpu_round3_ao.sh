#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
S="8192x8192x64 8192x8192x128 16384x8192x64 8192x4096x64 4096x4096x64 16384x4096x128"
for rep in 1 2; do for so in libmi355cube.so variants/libmi355cube_nt1.so variants/libmi355cube_nt0.so; do
  echo "== $so cold"; MI355CUBE_LIB=$PWD/cubecl_amd/csrc/$so timeout 300 python tools/ab_algos.py --rounds 5 --algos lp128 $S 2>&1 | tail -6
  echo "== $so warm"; MI355CUBE_LIB=$PWD/cubecl_amd/csrc/$so timeout 300 python tools/ab_algos.py --rounds 5 --warm --algos lp128 $S 2>&1 | tail -6
done; done 2>&1 | tee gpurun_out/r03ao_nt_c.txt
