#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
S="1536x1536x4096 2048x1536x4096 1792x2048x8192 2560x1024x2048 3200x1024x2048 768x3072x14336 1536x1536x8192 2048x1536x2048 1536x2048x16384 1024x1536x4096 1280x1024x4096 640x1024x8192 1792x1024x4096"
for rep in 1 2; do for so in libmi355cube.so variants/libmi355cube_floor1.so variants/libmi355cube_floor0.so variants/libmi355cube_ceil0.so; do
  echo "== $so cold"; MI355CUBE_LIB=$PWD/cubecl_amd/csrc/$so timeout 300 python tools/ab_algos.py --rounds 5 --algos lp128 $S 2>&1 | tail -13
done; done 2>&1 | tee gpurun_out/r03al_floor.txt
