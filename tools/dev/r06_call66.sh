set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export AUDIT_C32=1 AUDIT_ALL_TIMES=1
for seed in 2001 2002 2003 2004; do timeout 1200 python tools/dev/random_audit.py $seed 96; done > gpurun_out/r06_random_audit_c32.txt 2>&1
echo "bf16 -> f32 C: $(grep -c 'AUTO ->' gpurun_out/r06_random_audit_c32.txt) cases, $(grep -c BEHIND gpurun_out/r06_random_audit_c32.txt) behind"
awk '/^== rhs/{lay=$3} /BEHIND/{print lay, $0}' gpurun_out/r06_random_audit_c32.txt | cut -c1-260
