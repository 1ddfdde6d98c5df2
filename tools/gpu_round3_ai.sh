#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
S="128x256x8192 512x1024x2048 96x96x16384 1024x1024x4096 512x512x8192 128x8192x8192 1024x1536x4096 6144x6144x6144 256x2048x8192 64x28672x8192 2048x1024x4096"
for rep in 1 2; do for so in libmi355cube.so variants/libmi355cube_oldfold.so; do
  echo "== $so cold"; MI355CUBE_LIB=$PWD/cubecl_amd/csrc/$so timeout 300 python tools/ab_algos.py --rounds 5 --algos auto $S 2>&1 | tail -11
done; done 2>&1 | tee gpurun_out/r03ai_fold_ab.txt
