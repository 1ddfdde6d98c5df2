cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
SH="1x8192x8192 16x8192x8192 64x8192x8192 16x28672x8192 64x28672x8192 128x28672x8192 32x4096x4096 8x57344x4096 4x4096x14336 48x14336x4096"
{ echo "== NN (row-major rhs [K][N]), few rows: AUTO against the kernels forced";
  timeout 600 python tools/ab_algos.py --nn --rounds 3 --algos auto,lp128 $SH 2>&1 | tail -12;
  echo "== the same shapes, rhs stored [N][K] (NT)";
  timeout 600 python tools/ab_algos.py --rounds 3 --algos auto,lp128 $SH 2>&1 | tail -12; } > gpurun_out/r03_nn_few_rows_after.txt 2>&1
cat gpurun_out/r03_nn_few_rows_after.txt
timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_select_audit.py tests/test_gpu_gemm_fuzz.py tests/test_gpu_full_size.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -5
