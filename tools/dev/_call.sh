set -x
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_select_audit.py -q -x -m gpu -k "f32 or stream64 or audit or capture" 2>&1 | tail -8
timeout 900 python tools/ab_algos.py --f32 --rounds 5 --algos auto,skinny,stream64 \
  1x8192x8192 2x8192x8192 3x8192x8192 4x8192x8192 5x8192x8192 6x8192x8192 8x8192x8192 1x4096x4096 2x4096x4096 4x4096x4096 8x4096x4096 4x28672x4096 8x28672x4096 \
  8x2048x8192 4x2048x2048 8x1024x1024 8192x4x8192 8192x8x8192 4096x8x4096 2x16384x2048 8x16384x2048 16x6144x6144 16x5120x8192 16x28672x4096 > gpurun_out/r05_stream64_f32_few_rows.txt 2>&1
cat gpurun_out/r05_stream64_f32_few_rows.txt
