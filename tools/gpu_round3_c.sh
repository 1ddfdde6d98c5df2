#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_gemm.py -m gpu -q --no-header --timeout 300 -p no:cacheprovider --maxfail=10 -k "lp256x128" > $OUT/r03c_pytest.log 2>&1
echo "pytest exit $?"; tail -n 15 $OUT/r03c_pytest.log
timeout 600 python tools/ab_algos.py --rounds 5 2048x2048x2048 2048x2048x8192 4096x2048x2048 4096x2048x4096 2048x4096x4096 3072x3072x3072 4096x4096x1024 1024x4096x4096 2560x2560x2560 4096x3072x2048 1024x1024x4096 > $OUT/r03c_ab.txt 2>&1
cat $OUT/r03c_ab.txt
