set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export AUDIT_ALL_TIMES=1
for seed in 701 702 703 704 705 706 707 708; do timeout 1500 python tools/dev/random_audit.py $seed 96; done > gpurun_out/r06_audit_times_fit.txt 2>&1
for seed in 801 802 803 804 601 602 603 604; do timeout 1500 python tools/dev/random_audit.py $seed 96; done > gpurun_out/r06_audit_times_held_out.txt 2>&1
grep -c "AUTO ->" gpurun_out/r06_audit_times_fit.txt gpurun_out/r06_audit_times_held_out.txt
