set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python tools/dev/tiny_ta_probe.py > gpurun_out/r06_tiny_relayout.txt 2>&1; cat gpurun_out/r06_tiny_relayout.txt
timeout 1500 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_gemm_fuzz.py tests/test_gpu_layout_reduce_fuzz.py tests/test_gpu_contiguous.py -q -m gpu --no-header -p no:cacheprovider 2>&1 | tail -5
