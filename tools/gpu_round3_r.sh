#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_full_size.py -m gpu -q --no-header -p no:cacheprovider -k "256x128" 2>&1 | tail -3
S="4096x2048x4096 2048x4096x4096 4096x2048x2048 2560x2560x4096 4096x1536x8192 3072x2560x1024 2048x2048x2048 4096x2048x8192"
for rep in 1 2; do for so in libmi355cube.so variants/libmi355cube_nohyb.so; do
  echo "== $so"; MI355CUBE_LIB=$PWD/cubecl_amd/csrc/$so timeout 300 python tools/ab_algos.py --rounds 5 --algos lp128,lp256x128 $S 2>&1 | tail -8
done; done 2>&1 | tee gpurun_out/r03r_hyb.txt
