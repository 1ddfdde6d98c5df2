#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
S="64x8192x8192 8192x64x8192 16x8192x8192 32x8192x8192 16x28672x8192 32x14336x4096 48x4096x4096 64x14336x4096 32x512x8192"
MI355CUBE_LIB=$PWD/cubecl_amd/csrc/variants/libmi355cube_nlb4.so timeout 600 python -m pytest tests/test_gpu_gemm.py -m gpu -q --no-header -p no:cacheprovider -k "stream64" 2>&1 | tail -3
for rep in 1 2; do for so in libmi355cube.so variants/libmi355cube_nlb4.so; do
  echo "== $so cold"; MI355CUBE_LIB=$PWD/cubecl_amd/csrc/$so timeout 300 python tools/ab_algos.py --rounds 5 --algos stream64 $S 2>&1 | tail -9
  echo "== $so warm"; MI355CUBE_LIB=$PWD/cubecl_amd/csrc/$so timeout 300 python tools/ab_algos.py --rounds 5 --warm --algos stream64 $S 2>&1 | tail -9
done; done 2>&1 | tee gpurun_out/r03q_nlb4.txt
