set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for seed in 701 702 703 704 705 706 707 708; do timeout 1500 python tools/dev/random_audit.py $seed 96; done > gpurun_out/r06_random_audit_fresh_seeds_after.txt 2>&1
grep -c "AUTO ->" gpurun_out/r06_random_audit_fresh_seeds_after.txt; grep "BEHIND" gpurun_out/r06_random_audit_fresh_seeds_after.txt
echo "== held out"
for seed in 801 802 803 804; do timeout 1500 python tools/dev/random_audit.py $seed 96; done > gpurun_out/r06_random_audit_held_out.txt 2>&1
grep -c "AUTO ->" gpurun_out/r06_random_audit_held_out.txt; grep "BEHIND" gpurun_out/r06_random_audit_held_out.txt
