set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
echo "== [N][K] rhs"
timeout 1500 python tools/ab_algos.py --rounds 5 --algos auto,lp256x192,lp192x192,lp256qm 8192x4352x4096 8192x4352x8192 8192x4352x2048 8192x4608x4096 4352x8192x4096 8704x8192x4096 8192x8704x8192 12288x4352x4096 5120x5376x4096 4096x4608x6144
echo "== row-major rhs"
timeout 1500 python tools/ab_algos.py --nn --rounds 5 --algos auto,lp256x192,lp192x192,lp256qm 8192x4352x4096 8192x4608x4096 8704x8192x4096
} > gpurun_out/r06_tail_split_rule_ab2.txt 2>&1
cat gpurun_out/r06_tail_split_rule_ab2.txt
for seed in 601 602 603 604; do timeout 1500 python tools/dev/random_audit.py $seed 96; done > gpurun_out/r06_random_audit_tail.txt 2>&1
grep -c "AUTO ->" gpurun_out/r06_random_audit_tail.txt; grep "BEHIND" gpurun_out/r06_random_audit_tail.txt
timeout 1500 python -m pytest tests -m gpu -q --no-header --timeout 300 -p no:cacheprovider --maxfail=60 > gpurun_out/r06_pytest_tail.log 2>&1
echo "pytest exit $?"; grep -E "passed|failed" gpurun_out/r06_pytest_tail.log | tail -3; grep -E "^FAILED|^ERROR" gpurun_out/r06_pytest_tail.log | head -20
