set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export AUDIT_TA=1 AUDIT_ALL_TIMES=1
for seed in 4001 4002 4003 4004; do timeout 900 python tools/dev/random_audit.py $seed 96; done > gpurun_out/r06_random_audit_ta_after.txt 2>&1
echo "lhs [K][M], seeds of the rule: $(grep -c 'AUTO ->' gpurun_out/r06_random_audit_ta_after.txt) cases, $(grep -c BEHIND gpurun_out/r06_random_audit_ta_after.txt) behind"
for seed in 4101 4102 4103 4104; do timeout 900 python tools/dev/random_audit.py $seed 96; done > gpurun_out/r06_random_audit_ta_unseen.txt 2>&1
echo "lhs [K][M], unseen seeds: $(grep -c 'AUTO ->' gpurun_out/r06_random_audit_ta_unseen.txt) cases, $(grep -c BEHIND gpurun_out/r06_random_audit_ta_unseen.txt) behind"
awk '/BEHIND/{print}' gpurun_out/r06_random_audit_ta_after.txt gpurun_out/r06_random_audit_ta_unseen.txt | cut -c1-200 | head -40
export AUDIT_C32=1
for seed in 4201 4202; do timeout 900 python tools/dev/random_audit.py $seed 96; done > gpurun_out/r06_random_audit_ta_c32.txt 2>&1
echo "lhs [K][M], f32 C: $(grep -c 'AUTO ->' gpurun_out/r06_random_audit_ta_c32.txt) cases, $(grep -c BEHIND gpurun_out/r06_random_audit_ta_c32.txt) behind"
awk '/BEHIND/{print}' gpurun_out/r06_random_audit_ta_c32.txt | cut -c1-200 | head -20
unset AUDIT_TA AUDIT_ALL_TIMES AUDIT_C32
timeout 900 python -m pytest tests -q -m gpu --no-header -p no:cacheprovider -k "transposed or select or permuted or layout or contiguous" 2>&1 | tail -5
