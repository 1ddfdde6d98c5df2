#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests/test_gpu_gemm.py -m gpu -q --no-header -p no:cacheprovider -k "lp128 or 256x128 or row_major" 2>&1 | tail -3
MI355CUBE_LIB=$PWD/cubecl_amd/csrc/variants/libmi355cube_full4.so timeout 600 python -m pytest tests/test_gpu_gemm.py -m gpu -q --no-header -p no:cacheprovider -k "lp128 or row_major" 2>&1 | tail -3
S="128x28672x8192 96x28672x8192 128x14336x4096 128x8192x8192 2048x2048x2048 1024x4096x4096 2048x2048x8192 96x57344x4096 1024x1024x4096 128x57344x4096"
for rep in 1 2; do for so in libmi355cube.so variants/libmi355cube_full4.so; do
  echo "== $so cold"; MI355CUBE_LIB=$PWD/cubecl_amd/csrc/$so timeout 300 python tools/ab_algos.py --rounds 5 --algos lp128 $S 2>&1 | tail -10
done; done 2>&1 | tee gpurun_out/r03z_full4.txt
