set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export AUDIT_C32=1 AUDIT_ALL_TIMES=1
for seed in 2001 2002 2003 2004; do timeout 1200 python tools/dev/random_audit.py $seed 96; done > gpurun_out/r06_random_audit_c32_after.txt 2>&1
echo "bf16 -> f32 C, seeds of the change: $(grep -c 'AUTO ->' gpurun_out/r06_random_audit_c32_after.txt) cases, $(grep -c BEHIND gpurun_out/r06_random_audit_c32_after.txt) behind"
for seed in 2101 2102 2103 2104; do timeout 1200 python tools/dev/random_audit.py $seed 96; done > gpurun_out/r06_random_audit_c32_held_out.txt 2>&1
echo "bf16 -> f32 C, unseen seeds: $(grep -c 'AUTO ->' gpurun_out/r06_random_audit_c32_held_out.txt) cases, $(grep -c BEHIND gpurun_out/r06_random_audit_c32_held_out.txt) behind"
awk '/^== rhs/{lay=$3} /BEHIND/{print lay, $0}' gpurun_out/r06_random_audit_c32_after.txt gpurun_out/r06_random_audit_c32_held_out.txt | cut -c1-270
timeout 1500 python tools/dev/batched_audit.py > gpurun_out/r06_batched_audit_c32.txt 2>&1; tail -1 gpurun_out/r06_batched_audit_c32.txt; grep BEHIND gpurun_out/r06_batched_audit_c32.txt | cut -c1-300
timeout 600 python tools/dev/batched_audit.py 512x2048x2048x2048 64x2048x2048x2048 2>&1 | cut -c1-300
