#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
S="64x8192x8192 8192x64x8192 16x8192x8192 32x8192x8192 4x8192x8192 16x28672x8192 32x14336x4096 48x4096x4096 64x14336x4096 8x8192x8192 32x512x8192 64x2048x8192"
for rep in 1 2; do for so in libmi355cube.so variants/libmi355cube_snt.so; do
  echo "== $so cold"; MI355CUBE_LIB=$PWD/cubecl_amd/csrc/$so timeout 300 python tools/ab_algos.py --rounds 5 --algos stream64 $S 2>&1 | tail -12
  echo "== $so warm"; MI355CUBE_LIB=$PWD/cubecl_amd/csrc/$so timeout 300 python tools/ab_algos.py --rounds 5 --warm --algos stream64 $S 2>&1 | tail -12
done; done 2>&1 | tee $OUT/r03n_stream_nt.txt
