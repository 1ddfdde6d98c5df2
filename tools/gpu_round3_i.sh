#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
S="64x8192x8192 16x8192x8192 32x14336x4096"
for rep in 1 2; do for so in cubecl_amd/csrc/libmi355cube.so cubecl_amd/csrc/variants/libmi355cube_abl1.so cubecl_amd/csrc/variants/libmi355cube_abl2.so cubecl_amd/csrc/variants/libmi355cube_noshift.so; do
  echo "== $so cold"; MI355CUBE_LIB=$PWD/$so timeout 300 python tools/ab_algos.py --rounds 5 --algos stream64 $S 2>&1 | tail -3
  echo "== $so warm"; MI355CUBE_LIB=$PWD/$so timeout 300 python tools/ab_algos.py --rounds 5 --warm --algos stream64 $S 2>&1 | tail -3
done; done 2>&1 | tee $OUT/r03i_s64_abl.txt
