set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python tools/ab_algos.py --f32 --rounds 3 --algos auto,f32,stream64,lp256w4 64x64x65536 64x128x16384 32x256x32768 64x512x8192 64x1024x8192 16x64x32768 48x2048x16384 64x65536x512 64x32768x1024 64x49152x2048 32x65536x512 16x131072x256 64x40960x4096 8x128x65536 > gpurun_out/r06_f32_stream_corners_ab.txt 2>&1
cat gpurun_out/r06_f32_stream_corners_ab.txt
timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_runtime.py -q --no-header -p no:cacheprovider -x -k "stays_fused or all_reduce or exchange or collective or comm" --timeout 300 2>&1 | tail -5
