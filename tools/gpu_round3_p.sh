#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2 3; do for so in libmi355cube.so variants/libmi355cube_nont.so; do
  echo "== $so"; MI355CUBE_LIB=$PWD/cubecl_amd/csrc/$so timeout 300 python tools/ab_algos.py --rounds 5 --algos lp128 64x28672x8192 128x28672x8192 100x57344x4096 28672x128x8192 96x32768x4096 2>&1 | tail -5
done; done 2>&1 | tee gpurun_out/r03p_lp128_nt_ab.txt
