set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
for off in 9000 10000 11000 12000; do
  MI355_FUZZ_OFFSET=$off timeout 900 python -m pytest tests/test_gpu_gemm_fuzz.py -q -m gpu --no-header -p no:cacheprovider 2>&1 | tail -2
done
} > gpurun_out/r06_soak_end.txt 2>&1
cat gpurun_out/r06_soak_end.txt
