#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
S="64x8192x8192 8192x64x8192 16x8192x8192 32x8192x8192 4x8192x8192 1x8192x8192 2x8192x8192 16x28672x8192 64x28672x8192 128x28672x8192 128x8192x8192 32x14336x4096 8192x8192x64 8192x8192x128 16384x8192x64"
echo "== cold (rotating operand sets)"; timeout 900 python tools/ab_algos.py --rounds 5 --algos auto,stream64,lp128,skinny $S 2>&1 | tee $OUT/r03e_cold.txt
echo "== warm (one operand set)"; timeout 900 python tools/ab_algos.py --rounds 5 --warm --algos auto,stream64,lp128,skinny $S 2>&1 | tee $OUT/r03e_warm.txt
