set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export AUDIT_FP8=1 AUDIT_ALL_TIMES=1
for seed in 3001 3002 3003 3004; do timeout 600 python tools/dev/random_audit.py $seed 96; done > gpurun_out/r06_random_audit_fp8_after.txt 2>&1
echo "fp8, seeds of the rule: $(grep -c 'AUTO ->' gpurun_out/r06_random_audit_fp8_after.txt) cases, $(grep -c BEHIND gpurun_out/r06_random_audit_fp8_after.txt) behind"
for seed in 3101 3102 3103 3104; do timeout 600 python tools/dev/random_audit.py $seed 96; done > gpurun_out/r06_random_audit_fp8_unseen.txt 2>&1
echo "fp8, unseen seeds: $(grep -c 'AUTO ->' gpurun_out/r06_random_audit_fp8_unseen.txt) cases, $(grep -c BEHIND gpurun_out/r06_random_audit_fp8_unseen.txt) behind"
awk '/^== rhs/{lay=$3} /BEHIND/{print lay, $0}' gpurun_out/r06_random_audit_fp8_after.txt gpurun_out/r06_random_audit_fp8_unseen.txt | cut -c1-200
unset AUDIT_FP8 AUDIT_ALL_TIMES
timeout 900 python -m pytest tests -q -m gpu --no-header -p no:cacheprovider -k "fp8 or f8 or scaled or select" 2>&1 | tail -3
