set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export AUDIT_F32=1
for seed in 1001 1002 1003 1004; do timeout 900 python tools/dev/random_audit.py $seed 96; done > gpurun_out/r06_random_audit_f32_after2.txt 2>&1
echo "fit seeds: $(grep -c 'AUTO ->' gpurun_out/r06_random_audit_f32_after2.txt) cases, $(grep -c BEHIND gpurun_out/r06_random_audit_f32_after2.txt) behind"
for seed in 1301 1302 1303 1304; do timeout 900 python tools/dev/random_audit.py $seed 96; done > gpurun_out/r06_random_audit_f32_held_out2.txt 2>&1
echo "fresh seeds: $(grep -c 'AUTO ->' gpurun_out/r06_random_audit_f32_held_out2.txt) cases, $(grep -c BEHIND gpurun_out/r06_random_audit_f32_held_out2.txt) behind"
awk '/^== rhs/{lay=$3} /BEHIND/{print lay, $0}' gpurun_out/r06_random_audit_f32_after2.txt gpurun_out/r06_random_audit_f32_held_out2.txt | cut -c1-150
unset AUDIT_F32
timeout 900 python tools/ab_algos.py --f32 --rounds 5 --algos auto,f32,lp256w4 4096x4096x4096 2048x2048x2048 3072x3072x3072 5120x5120x5120 6144x6144x6144 8192x8192x8192 4160x4096x4096 4672x3968x4096 > gpurun_out/r06_f32_squares_ab.txt 2>&1; tail -12 gpurun_out/r06_f32_squares_ab.txt
timeout 900 python -m pytest tests/test_gpu_select_audit.py tests/test_gpu_gemm.py -q -m gpu --no-header -p no:cacheprovider -k "f32 or select or audit" 2>&1 | tail -4
