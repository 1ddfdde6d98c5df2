#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
S="2048x2048x2048 1024x4096x4096 2048x2048x8192 4096x2048x2048 1536x1536x4096 1024x1024x4096 3072x2560x1024"
for rep in 1 2; do for so in libmi355cube.so variants/libmi355cube_kst1.so variants/libmi355cube_kst2.so variants/libmi355cube_kst3.so; do
  echo "== $so cold"; MI355CUBE_LIB=$PWD/cubecl_amd/csrc/$so timeout 300 python tools/ab_algos.py --rounds 5 --algos lp128 $S 2>&1 | tail -7
done; done 2>&1 | tee gpurun_out/r03aa_kstag.txt
