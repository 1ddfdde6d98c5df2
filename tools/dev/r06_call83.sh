set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_reduce.py tests/test_gpu_layout_reduce_fuzz.py -q -m gpu --no-header -p no:cacheprovider -x 2>&1 | tail -3 | cut -c1-250
for seed in 11 12; do timeout 600 python tools/dev/reduce_audit.py $seed 40; done > gpurun_out/r06_reduce_audit_after.txt 2>&1
grep "under 0.35" gpurun_out/r06_reduce_audit_after.txt
grep " 48382\| 199410\| 100576\| 12997\| 635518\| 1404,\|  13,\|   4," gpurun_out/r06_reduce_audit_after.txt | cut -c1-150
timeout 300 python tools/axis_probe.py quick 2>&1 | tail -12 | cut -c1-200
