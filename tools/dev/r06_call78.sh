set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
AUDIT_TA=1 AUDIT_ALL_TIMES=1 timeout 1500 python tools/dev/batched_audit.py > gpurun_out/r06_batched_audit_ta.txt 2>&1; tail -1 gpurun_out/r06_batched_audit_ta.txt; grep BEHIND gpurun_out/r06_batched_audit_ta.txt | cut -c1-250
grep -c "nothing to compare" gpurun_out/r06_batched_audit_ta.txt
