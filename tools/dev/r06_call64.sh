set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export AUDIT_F32=1
for seed in 1001 1002 1003 1004; do timeout 900 python tools/dev/random_audit.py $seed 96; done > gpurun_out/r06_random_audit_f32_after.txt 2>&1
echo "fit seeds: $(grep -c 'AUTO ->' gpurun_out/r06_random_audit_f32_after.txt) cases, $(grep -c BEHIND gpurun_out/r06_random_audit_f32_after.txt) behind"
for seed in 1201 1202 1203 1204; do timeout 900 python tools/dev/random_audit.py $seed 96; done > gpurun_out/r06_random_audit_f32_held_out.txt 2>&1
echo "held-out seeds: $(grep -c 'AUTO ->' gpurun_out/r06_random_audit_f32_held_out.txt) cases, $(grep -c BEHIND gpurun_out/r06_random_audit_f32_held_out.txt) behind"
awk '/^== rhs/{lay=$3} /BEHIND/{print lay, $0}' gpurun_out/r06_random_audit_f32_held_out.txt | cut -c1-150
unset AUDIT_F32
timeout 900 python tools/ab_algos.py --f32 --rounds 5 --algos auto,f32,lp256w4 4096x4096x4096 2048x2048x2048 3072x3072x3072 5120x5120x5120 6144x6144x6144 8192x8192x8192 4096x4096x512 1024x1024x1024 1536x1536x1536 2560x2560x2560
timeout 900 python -m pytest tests/test_gpu_select_audit.py tests/test_gpu_gemm.py -q -m gpu --no-header -p no:cacheprovider -k "f32 or select or audit" 2>&1 | tail -4
