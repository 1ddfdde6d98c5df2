#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
S=""
for mn in 4096x2048 2048x4096 3072x2560 3584x2048 2304x3584 4096x1536 3072x2048 2560x2560 3072x3072 4096x1792; do for k in 1024 2048 3072 4096 8192; do S="$S ${mn}x${k}"; done; done
timeout 1200 python tools/ab_algos.py --rounds 5 --algos lp128,lp256x128,lp256w4 $S > $OUT/r03d_ab.txt 2>&1
cat $OUT/r03d_ab.txt
