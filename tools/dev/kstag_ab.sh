cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
SH="64x28672x8192 128x28672x8192 96x14336x4096 64x57344x4096 128x8192x8192 16x28672x8192 2048x2048x2048 1024x4096x4096 256x2048x8192 512x512x8192 8192x8192x64 1536x2048x4096"
{ for rep in 1 2; do
    for v in "" kstag1 kstag3; do
      so=${v:+$PWD/cubecl_amd/csrc/variants/libmi355cube_$v.so}
      echo "== ${v:-product} NT lp128"; env ${so:+MI355CUBE_LIB=$so} timeout 300 python tools/ab_algos.py --rounds 3 --algos lp128 $SH 2>&1 | tail -12
    done
  done
  for v in "" kstag3; do
    so=${v:+$PWD/cubecl_amd/csrc/variants/libmi355cube_$v.so}
    echo "== ${v:-product} NN lp128"; env ${so:+MI355CUBE_LIB=$so} timeout 300 python tools/ab_algos.py --nn --rounds 3 --algos lp128 $SH 2>&1 | tail -12
  done; } > gpurun_out/r03_lp128_k_stagger.txt 2>&1
