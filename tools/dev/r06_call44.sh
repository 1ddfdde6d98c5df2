set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_gemm.py -q -m gpu --no-header -p no:cacheprovider -k "tail or partly or leftover or strip" 2>&1 | tail -3
for v in "" _tailold "" _tailold; do
  so=cubecl_amd/csrc/libmi355cube.so; [ -n "$v" ] && so=cubecl_amd/csrc/variants/libmi355cube$v.so
  echo "== $so"
  MI355CUBE_LIB=$PWD/$so timeout 600 python tools/ab_algos.py --rounds 5 --algos auto,lp256qm 4608x4096x8192 4864x4096x8192 8448x8192x8192 4096x4352x8192 6144x6400x8192 8192x8448x4096
  MI355CUBE_LIB=$PWD/$so timeout 600 python tools/ab_algos.py --nn --rounds 5 --algos auto,lp256qm 4608x4096x8192 6144x6144x6144
done
} > gpurun_out/r06_tail_split_main_on_qm.txt 2>&1
cat gpurun_out/r06_tail_split_main_on_qm.txt
