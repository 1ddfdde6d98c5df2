#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
S="64x28672x8192 64x14336x4096 48x28672x4096 40x16384x8192 64x32768x4096"
for rep in 1 2; do for so in libmi355cube.so variants/libmi355cube_u1.so variants/libmi355cube_u2.so variants/libmi355cube_u3.so variants/libmi355cube_u4.so variants/libmi355cube_u5.so; do
  echo "== $so cold"; MI355CUBE_LIB=$PWD/cubecl_amd/csrc/$so timeout 300 python tools/ab_algos.py --rounds 5 --algos stream64,lp128 $S 2>&1 | tail -5
done; done 2>&1 | tee gpurun_out/r03x_geom_two2.txt
