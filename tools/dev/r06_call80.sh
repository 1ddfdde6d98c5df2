set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
MI355_FUZZ_OFFSET=9000 timeout 300 python -m pytest "tests/test_gpu_gemm_fuzz.py::test_auto_dispatch_on_random_descriptors[90]" -q -m gpu --no-header -p no:cacheprovider 2>&1 | tail -40
MI355_FUZZ_OFFSET=12000 timeout 300 python -m pytest "tests/test_gpu_gemm_fuzz.py::test_auto_dispatch_on_random_descriptors[149]" -q -m gpu --no-header -p no:cacheprovider 2>&1 | tail -25
