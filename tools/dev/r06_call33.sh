set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_select_audit.py tests/test_gpu_gemm.py tests/test_gpu_gemm_fuzz.py tests/test_gpu_full_size.py -m gpu -q --no-header --timeout 600 -p no:cacheprovider --maxfail=30 > gpurun_out/r06b_gemm_pytest.log 2>&1
echo "pytest exit $?"; grep -E "passed|failed" gpurun_out/r06b_gemm_pytest.log | tail -3; grep -E "^FAILED|^ERROR" gpurun_out/r06b_gemm_pytest.log | head -30
for seed in 601 602 603 604; do timeout 1500 python tools/dev/random_audit.py $seed 96; done > gpurun_out/r06_random_audit_qm2.txt 2>&1
grep -c "AUTO ->" gpurun_out/r06_random_audit_qm2.txt; grep "BEHIND" gpurun_out/r06_random_audit_qm2.txt
tail -3 gpurun_out/select_audit.txt gpurun_out/select_audit_nn.txt
