set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
for off in 9000 12000 13000 14000; do
  MI355_FUZZ_OFFSET=$off timeout 900 python -m pytest tests/test_gpu_gemm_fuzz.py -q -m gpu --no-header -p no:cacheprovider 2>&1 | tail -2
done
} > gpurun_out/r06_soak_end2.txt 2>&1
cat gpurun_out/r06_soak_end2.txt
for seed in 11 12; do timeout 600 python tools/dev/reduce_audit.py $seed 40; done > gpurun_out/r06_reduce_audit.txt 2>&1
grep -c "of HBM" gpurun_out/r06_reduce_audit.txt; grep "SLOW\|under 0.35" gpurun_out/r06_reduce_audit.txt | cut -c1-200 | head -50
