set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_select_audit.py tests/test_gpu_gemm.py tests/test_gpu_gemm_fuzz.py -m gpu -q --no-header --timeout 600 -p no:cacheprovider --maxfail=30 > gpurun_out/r06b_gemm_pytest.log 2>&1
echo "pytest exit $?"; grep -E "passed|failed" gpurun_out/r06b_gemm_pytest.log | tail -3; grep -E "^FAILED|^ERROR" gpurun_out/r06b_gemm_pytest.log | head -30
{
timeout 900 python tools/ab_algos.py --rounds 5 --algos auto,lp256w4,lp256m16,lp256qm,lp256x192,lp192x192 4608x4096x8192 4864x4096x8192 4352x4096x4096 5120x4096x4096 8448x8192x8192 6144x6144x6144 4096x4096x1024 4096x4096x2048 4096x4096x8192 3584x3584x3584 4096x4096x512
timeout 600 python tools/ab_algos.py --nn --rounds 5 --algos auto,lp256w4,lp256q,lp256qm,lp256x192 8192x8192x448 8192x8192x384 4096x4096x4096 4096x4096x1024 6144x6144x6144 4608x4096x8192
} > gpurun_out/r06_qm_rule_ab.txt 2>&1
cat gpurun_out/r06_qm_rule_ab.txt
for seed in 601 602 603 604; do timeout 1500 python tools/dev/random_audit.py $seed 96; done > gpurun_out/r06_random_audit_qm2.txt 2>&1
grep -c "AUTO ->" gpurun_out/r06_random_audit_qm2.txt; grep "BEHIND" gpurun_out/r06_random_audit_qm2.txt
