#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp; R=$PWD
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03ag -o t -- python $R/tools/ab_algos.py --rounds 2 --algos lp128 128x256x8192 512x1024x2048 96x96x16384 > $R/gpurun_out/r03ag.log 2>&1 )
cat gpurun_out/r03ag.log | tail -4
f=$(find gpurun_out/r03ag -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-200
