set -x
cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_select_audit.py tests/test_gpu_full_size.py -q -m gpu -n 1 2>&1 | tail -15
timeout 700 python tools/ab_algos.py --nn --rounds 5 --algos auto,lp128,lp256x128,lp256w4,lp256x192,lp192x192 \
  23664x352x14336 13096x128x512 16384x512x1024 32768x256x1024 1024x1024x8192 768x3072x14336 1536x2048x16384 6144x6144x1024 8192x4096x2048 \
  4096x4096x1024 5120x4096x4096 7168x4096x2048 2048x8192x1024 12288x1024x1024 1280x1280x2048 1792x1792x1024 640x24576x1024 4096x1792x1024 \
  3840x3840x768 4608x2816x640 > gpurun_out/r05_tile_nn_ab2.txt 2>&1
tail -30 gpurun_out/r05_tile_nn_ab2.txt
timeout 900 python tools/dev/random_audit.py 11 40 > gpurun_out/r05_random_audit_nn_s11.txt 2>&1
grep -c . gpurun_out/r05_random_audit_nn_s11.txt; grep -i "behind\|==" gpurun_out/r05_random_audit_nn_s11.txt | tail -30
