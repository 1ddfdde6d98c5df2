set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
for off in 15000 16000 17000 18000; do
  MI355_FUZZ_OFFSET=$off timeout 900 python -m pytest tests/test_gpu_gemm_fuzz.py tests/test_gpu_layout_reduce_fuzz.py -q -m gpu --no-header -p no:cacheprovider 2>&1 | grep -a " passed\| failed\|FAILED" | head -4
done
for rep in 1 2 3; do
  timeout 600 python -m pytest tests/test_gpu_reduce.py -q -m gpu --no-header -p no:cacheprovider 2>&1 | grep -a " passed\| failed\|FAILED" | head -3
done
} > gpurun_out/r06_soak_last.txt 2>&1
cat gpurun_out/r06_soak_last.txt
