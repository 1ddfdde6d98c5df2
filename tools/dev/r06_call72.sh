set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export AUDIT_TA=1 AUDIT_ALL_TIMES=1
for seed in 4001 4002 4003 4004; do timeout 900 python tools/dev/random_audit.py $seed 96; done > gpurun_out/r06_random_audit_ta.txt 2>&1
echo "lhs [K][M] x row-major rhs: $(grep -c 'AUTO ->' gpurun_out/r06_random_audit_ta.txt) cases, $(grep -c BEHIND gpurun_out/r06_random_audit_ta.txt) behind"
awk '/BEHIND/{print}' gpurun_out/r06_random_audit_ta.txt | cut -c1-200 | head -60
grep -i "error\|Traceback" gpurun_out/r06_random_audit_ta.txt | head -5
