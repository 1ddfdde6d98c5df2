#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
S="64x8192x8192 16x8192x8192 32x8192x8192 4x8192x8192 48x4096x4096 32x512x8192 64x2048x8192"
for rep in 1 2; do for so in libmi355cube.so variants/libmi355cube_v1.so variants/libmi355cube_v2.so variants/libmi355cube_v3.so variants/libmi355cube_v4.so; do
  echo "== $so cold"; MI355CUBE_LIB=$PWD/cubecl_amd/csrc/$so timeout 300 python tools/ab_algos.py --rounds 5 --algos stream64 $S 2>&1 | tail -7
  echo "== $so warm"; MI355CUBE_LIB=$PWD/cubecl_amd/csrc/$so timeout 300 python tools/ab_algos.py --rounds 5 --warm --algos stream64 $S 2>&1 | tail -7
done; done 2>&1 | tee gpurun_out/r03v_geom2.txt
