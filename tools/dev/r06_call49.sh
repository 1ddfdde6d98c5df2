set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --no-header --timeout 300 -p no:cacheprovider --maxfail=60 > gpurun_out/r06_pytest.log 2>&1
echo "pytest exit $?"; grep -E "passed|failed" gpurun_out/r06_pytest.log | tail -3; grep -E "^FAILED|^ERROR" gpurun_out/r06_pytest.log | head -20
for seed in 601 602 603 604; do timeout 1500 python tools/dev/random_audit.py $seed 96; done > gpurun_out/r06_random_audit_final2.txt 2>&1
grep -c "AUTO ->" gpurun_out/r06_random_audit_final2.txt; grep "BEHIND" gpurun_out/r06_random_audit_final2.txt
