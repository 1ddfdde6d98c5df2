#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
S="1024x1024x4096 1024x1024x8192 1536x1536x4096 512x512x8192 768x768x16384 1024x512x8192 64x8192x8192 128x8192x8192 2048x512x8192 1024x2048x8192"
for rep in 1 2; do for so in libmi355cube.so variants/libmi355cube_want1.so; do
  echo "== $so cold"; MI355CUBE_LIB=$PWD/cubecl_amd/csrc/$so timeout 300 python tools/ab_algos.py --rounds 5 --algos lp128 $S 2>&1 | tail -10
done; done 2>&1 | tee gpurun_out/r03ab_want.txt
