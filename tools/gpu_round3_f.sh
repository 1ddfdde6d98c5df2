#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_select_audit.py tests/test_gpu_gemm.py tests/test_gpu_full_size.py -m gpu -q --no-header --timeout 600 -p no:cacheprovider --maxfail=10 -k "audit or select or 256x128" > $OUT/r03f_pytest.log 2>&1
echo "pytest exit $?"; tail -n 30 $OUT/r03f_pytest.log; cat $OUT/select_audit.txt
