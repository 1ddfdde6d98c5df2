set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_select_audit.py -q -m gpu --no-header -p no:cacheprovider 2>&1 | tail -25 | cut -c1-250
timeout 300 python tools/dev/tiny_ta_probe.py 2>&1 | head -13
