set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_gemm.py -q -m gpu --no-header -p no:cacheprovider -k "192 or narrow or tile or w4 or fp8 or scaled or f32" 2>&1 | tail -3
for v in "" _w4old "" _w4old; do
  so=cubecl_amd/csrc/libmi355cube.so; [ -n "$v" ] && so=cubecl_amd/csrc/variants/libmi355cube$v.so
  echo "== $so"
  MI355CUBE_LIB=$PWD/$so timeout 600 python tools/ab_algos.py --rounds 5 --algos lp256x192,lp192x192 3072x3072x3072 4096x2048x4096 2560x2560x3072 4096x3072x4096 3328x3328x4096 6144x6144x6144 8192x3072x2048
  MI355CUBE_LIB=$PWD/$so timeout 600 python tools/ab_algos.py --nn --rounds 5 --algos lp256x192,lp192x192 3072x3072x3072 4096x3072x4096
done
} > gpurun_out/r06_w4_slot_spread.txt 2>&1
cat gpurun_out/r06_w4_slot_spread.txt
