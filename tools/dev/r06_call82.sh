set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_reduce.py tests/test_gpu_layout_reduce_fuzz.py tests/test_gpu_full_size.py -q -m gpu --no-header -p no:cacheprovider -x 2>&1 | tail -25 | cut -c1-250
for seed in 11 12; do timeout 600 python tools/dev/reduce_audit.py $seed 40; done > gpurun_out/r06_reduce_audit_after.txt 2>&1
grep "under 0.35" gpurun_out/r06_reduce_audit_after.txt
grep "SLOW" gpurun_out/r06_reduce_audit_after.txt | sort -k8,8n | cut -c1-200 | head -60
{
for off in 9000 12000; do
  MI355_FUZZ_OFFSET=$off timeout 900 python -m pytest tests/test_gpu_gemm_fuzz.py -q -m gpu --no-header -p no:cacheprovider 2>&1 | tail -2
done
} > gpurun_out/r06_soak_end3.txt 2>&1
cat gpurun_out/r06_soak_end3.txt
