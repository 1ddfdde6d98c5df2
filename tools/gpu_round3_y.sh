#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
S=""
for r in 3 4 8 16 32 48 64; do for nk in 2048x4096 4096x4096 8192x2048 8192x8192 14336x4096 16384x8192 28672x4096 28672x8192 4096x14336 57344x4096 512x8192 1024x16384; do S="$S ${r}x${nk}"; done; done
timeout 1500 python tools/ab_algos.py --rounds 3 --algos auto,stream64,lp128,skinny $S > gpurun_out/r03y_fewrows.txt 2>&1
S2=""
for r in 96 128; do for nk in 4096x4096 8192x8192 14336x4096 28672x8192 57344x4096; do S2="$S2 ${r}x${nk}"; done; done
timeout 600 python tools/ab_algos.py --rounds 3 --algos auto,lp128,lp256w4 $S2 >> gpurun_out/r03y_fewrows.txt 2>&1
tail -100 gpurun_out/r03y_fewrows.txt
