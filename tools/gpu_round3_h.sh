#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_full_size.py -m gpu -q --no-header --timeout 300 -p no:cacheprovider --maxfail=10 -k "stream64 or skinny or benched" > $OUT/r03h_pytest.log 2>&1
echo "pytest exit $?"; tail -n 5 $OUT/r03h_pytest.log
S="64x8192x8192 8192x64x8192 16x8192x8192 32x8192x8192 4x8192x8192 16x28672x8192 32x14336x4096 48x4096x4096 64x14336x4096 8x8192x8192"
echo "== cold"; timeout 900 python tools/ab_algos.py --rounds 5 --algos auto,stream64,lp128 $S 2>&1 | tee $OUT/r03h_cold.txt
echo "== warm"; timeout 900 python tools/ab_algos.py --rounds 5 --warm --algos auto,stream64,lp128 $S 2>&1 | tee $OUT/r03h_warm.txt
