#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_full_size.py -m gpu -q --no-header -p no:cacheprovider -k "lp128 or split or skinny or benched or partly or tail or strip" 2>&1 | tail -3
S="128x256x8192 512x1024x2048 96x96x16384 1024x1024x4096 512x512x8192 128x8192x8192 64x1024x4096 1024x1536x4096 6144x6144x6144 4608x4096x8192 256x2048x8192"
timeout 600 python tools/ab_algos.py --rounds 5 --algos auto,lp128 $S 2>&1 | tee gpurun_out/r03ah_fold.txt
