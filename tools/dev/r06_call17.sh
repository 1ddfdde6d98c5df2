set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_gemm.py -q --no-header -p no:cacheprovider -x -k "stream64 or few_rows or skinny or 64" --timeout 600 2>&1 | tail -3
timeout 600 python tools/ab_algos.py --rounds 5 --algos stream64,lp128 64x8192x8192 8192x64x8192 48x8192x8192 32x8192x8192 16x8192x8192 64x7168x8192 64x8192x16384 16x28672x8192 > gpurun_out/r06_stream64_pitch33.txt 2>&1
cat gpurun_out/r06_stream64_pitch33.txt
