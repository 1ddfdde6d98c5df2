#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_full_size.py -m gpu -q --no-header -p no:cacheprovider -k "stream64 or benched" 2>&1 | tail -3
S="16x28672x8192 32x14336x4096 16x32000x4096 8x57344x4096 32x28672x2048 24x16384x8192"
for rep in 1 2; do for so in libmi355cube.so variants/libmi355cube_t1.so variants/libmi355cube_t2.so variants/libmi355cube_t3.so variants/libmi355cube_t4.so variants/libmi355cube_t5.so; do
  echo "== $so cold"; MI355CUBE_LIB=$PWD/cubecl_amd/csrc/$so timeout 300 python tools/ab_algos.py --rounds 5 --algos stream64 $S 2>&1 | tail -6
done; done 2>&1 | tee gpurun_out/r03w_geom_two.txt
