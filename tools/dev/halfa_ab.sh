cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
V=$PWD/cubecl_amd/csrc/variants/libmi355cube_halfa.so
SH="1x8192x8192 16x8192x8192 32x8192x8192 16x28672x8192 32x28672x8192 32x4096x4096 8x57344x4096 16x4096x14336 4x14336x4096 24x12288x4096"
{ MI355CUBE_LIB=$V timeout 300 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_gemm_fuzz.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -3
  for rep in 1 2; do
    for so in "" $V; do
      echo "== ${so:+halfa}${so:-product} NN lp128"; env ${so:+MI355CUBE_LIB=$so} timeout 300 python tools/ab_algos.py --nn --rounds 3 --algos lp128 $SH 2>&1 | tail -10
      echo "== ${so:+halfa}${so:-product} NT lp128"; env ${so:+MI355CUBE_LIB=$so} timeout 300 python tools/ab_algos.py --rounds 3 --algos lp128 $SH 2>&1 | tail -10
    done
  done; } > gpurun_out/r03_half_a_tile.txt 2>&1
