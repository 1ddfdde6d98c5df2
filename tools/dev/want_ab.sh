cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
SH="64x8192x8192 128x8192x8192 256x4096x8192 512x512x8192 1024x1024x4096 768x3072x14336 1024x512x8192 1536x2048x16384 256x2048x8192 2048x1024x4096 1024x1536x4096 384x384x4096 640x640x8192 896x1024x8192 1152x1024x4096 1280x1280x8192 1024x2048x16384 1792x1024x8192 96x4096x8192 192x8192x4096 2048x2048x2048 1408x1408x4096"
{ for rep in 1 2; do
    for v in "" oldwant; do
      so=${v:+$PWD/cubecl_amd/csrc/variants/libmi355cube_$v.so}
      echo "== ${v:-product} NT auto"; env ${so:+MI355CUBE_LIB=$so} timeout 300 python tools/ab_algos.py --rounds 3 --algos auto $SH 2>&1 | tail -22
    done
  done
  echo "== product NN auto"; timeout 300 python tools/ab_algos.py --nn --rounds 3 --algos auto $SH 2>&1 | tail -22
  timeout 600 python -m pytest tests/test_gpu_select_audit.py tests/test_gpu_gemm.py -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -3; } > gpurun_out/r03_split_target_rule.txt 2>&1
