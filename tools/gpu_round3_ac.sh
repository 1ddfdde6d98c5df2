#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests/test_gpu_gemm.py -m gpu -q --no-header -p no:cacheprovider -k "lp128 or split or skinny or select" 2>&1 | tail -3
S="1024x1024x4096 1024x1024x8192 1536x1536x4096 512x512x8192 768x768x16384 1024x512x8192 64x8192x8192 128x8192x8192 2048x512x8192 1024x2048x8192 256x256x16384 1024x1024x2048 2048x1024x4096 512x2048x4096 384x384x8192"
timeout 600 python tools/ab_algos.py --rounds 5 --algos auto,lp128,lp256w4,stream64 $S 2>&1 | tee gpurun_out/r03ac_split_fixed.txt
