set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export AUDIT_TA=1 AUDIT_ALL_TIMES=1 AUDIT_F32=1
for seed in 5001 5002 5003 5004; do timeout 900 python tools/dev/random_audit.py $seed 96; done > gpurun_out/r06_random_audit_ta_f32.txt 2>&1
echo "f32 lhs [K][M]: $(grep -c 'AUTO ->' gpurun_out/r06_random_audit_ta_f32.txt) cases, $(grep -c BEHIND gpurun_out/r06_random_audit_ta_f32.txt) behind"
awk '/BEHIND/{print}' gpurun_out/r06_random_audit_ta_f32.txt | cut -c1-200 | head -50
grep -c "no kernel takes it forced" gpurun_out/r06_random_audit_ta_f32.txt
grep "no kernel takes it forced" gpurun_out/r06_random_audit_ta_f32.txt | sort -k7,7nr | head -12
