set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_reduce.py tests/test_gpu_layout_reduce_fuzz.py tests/test_gpu_select_audit.py -q -m gpu --no-header -p no:cacheprovider 2>&1 | tail -12 | cut -c1-250
for seed in 11 12; do timeout 600 python tools/dev/reduce_audit.py $seed 40; done > gpurun_out/r06_reduce_audit_after.txt 2>&1
grep "under 0.35" gpurun_out/r06_reduce_audit_after.txt
