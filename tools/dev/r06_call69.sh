set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export AUDIT_C32=1 AUDIT_ALL_TIMES=1
for seed in 2101 2102 2103 2104; do timeout 1200 python tools/dev/random_audit.py $seed 96; done > gpurun_out/r06_random_audit_c32_held_out_after.txt 2>&1
echo "bf16 -> f32 C, second seeds: $(grep -c 'AUTO ->' gpurun_out/r06_random_audit_c32_held_out_after.txt) cases, $(grep -c BEHIND gpurun_out/r06_random_audit_c32_held_out_after.txt) behind"
for seed in 2201 2202 2203 2204; do timeout 1200 python tools/dev/random_audit.py $seed 96; done > gpurun_out/r06_random_audit_c32_unseen.txt 2>&1
echo "bf16 -> f32 C, unseen seeds: $(grep -c 'AUTO ->' gpurun_out/r06_random_audit_c32_unseen.txt) cases, $(grep -c BEHIND gpurun_out/r06_random_audit_c32_unseen.txt) behind"
awk '/^== rhs/{lay=$3} /BEHIND/{print lay, $0}' gpurun_out/r06_random_audit_c32_unseen.txt | cut -c1-270
timeout 1500 python tools/dev/batched_audit.py > gpurun_out/r06_batched_audit_c32_after.txt 2>&1; tail -1 gpurun_out/r06_batched_audit_c32_after.txt; grep BEHIND gpurun_out/r06_batched_audit_c32_after.txt | cut -c1-300
unset AUDIT_C32 AUDIT_ALL_TIMES
timeout 900 python -m pytest tests/test_gpu_select_audit.py tests/test_gpu_gemm.py -q -m gpu --no-header -p no:cacheprovider -k "select or audit or add" 2>&1 | tail -4
