#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
# (tiles of 128^2) x (K-tiles): 32x32 48x48 64x32 64x64 96x64 96x128 112x16 128x32 128x64 160x32 192x64 224x128 24x64 16x32 200x32 96x32 48x128
S="512x1024x2048 768x1024x3072 1024x1024x2048 1024x1024x4096 1024x1536x4096 1024x1536x8192 1792x1024x1024 2048x1024x2048 2048x1024x4096 2560x1024x2048 2048x1536x4096 1792x2048x8192 384x1024x4096 512x512x2048 3200x1024x2048 1536x1024x2048 768x1024x8192 384x3072x4096 2048x768x14336"
for rep in 1 2; do for so in libmi355cube.so variants/libmi355cube_rule0.so; do
  echo "== $so cold"; MI355CUBE_LIB=$PWD/cubecl_amd/csrc/$so timeout 300 python tools/ab_algos.py --rounds 5 --algos lp128 $S 2>&1 | tail -19
done; done 2>&1 | tee gpurun_out/r03ae_split_rule.txt
