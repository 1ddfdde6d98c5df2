set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
bash tools/gpu_check.sh r06 > gpurun_out/r06_gpu_check_final.log 2>&1
tail -n 30 gpurun_out/r06_gpu_check_final.log
{
for off in 5000 6000 7000 8000; do
  MI355_FUZZ_OFFSET=$off timeout 900 python -m pytest tests/test_gpu_gemm_fuzz.py -q -m gpu --no-header -p no:cacheprovider 2>&1 | tail -2
done
for rep in 1 2 3 4; do
  timeout 600 python -m pytest tests/test_gpu_gemm.py -q -m gpu --no-header -p no:cacheprovider -k "lp256qm or leftover_round_split or strip_split" 2>&1 | tail -1
done
timeout 600 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_layout_reduce_fuzz.py -q -m gpu --no-header -p no:cacheprovider 2>&1 | tail -1
} > gpurun_out/r06_soak_final.txt 2>&1
cat gpurun_out/r06_soak_final.txt
