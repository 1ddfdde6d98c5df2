set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
for off in 1000 2000 3000 4000; do
  MI355_FUZZ_OFFSET=$off timeout 900 python -m pytest tests/test_gpu_gemm_fuzz.py -q -m gpu --no-header -p no:cacheprovider 2>&1 | tail -2
done
for rep in 1 2 3 4 5 6 7 8; do
  timeout 600 python -m pytest tests/test_gpu_gemm.py -q -m gpu --no-header -p no:cacheprovider -k "lp256qm" 2>&1 | tail -1
done
timeout 600 python -m pytest tests/test_gpu_full_size.py -q -m gpu --no-header -p no:cacheprovider 2>&1 | tail -1
} > gpurun_out/r06_soak.txt 2>&1
cat gpurun_out/r06_soak.txt
