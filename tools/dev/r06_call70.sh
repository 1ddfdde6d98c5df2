set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export AUDIT_FP8=1 AUDIT_ALL_TIMES=1
for seed in 3001 3002 3003 3004; do timeout 600 python tools/dev/random_audit.py $seed 96; done > gpurun_out/r06_random_audit_fp8.txt 2>&1
echo "fp8 -> bf16 C: $(grep -c 'AUTO ->' gpurun_out/r06_random_audit_fp8.txt) cases, $(grep -c BEHIND gpurun_out/r06_random_audit_fp8.txt) behind"
awk '/^== rhs/{lay=$3} /BEHIND/{print lay, $0}' gpurun_out/r06_random_audit_fp8.txt | cut -c1-200 | head -70
grep -i "error\|Traceback" gpurun_out/r06_random_audit_fp8.txt | head -5
