#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_reduce.py tests/test_gpu_runtime.py tests/test_gpu_gemm.py -m gpu -q -s --no-header --timeout 300 -p no:cacheprovider --maxfail=10 -k "plane or max_lds or select" > $OUT/r03l_pytest.log 2>&1
echo "pytest exit $?"; grep "MODULE_LDS" $OUT/r03l_pytest.log; tail -n 6 $OUT/r03l_pytest.log
