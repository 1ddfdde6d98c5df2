set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_gemm.py -q -m gpu --no-header -p no:cacheprovider -k "tail or partly or leftover or strip" 2>&1 | tail -3
echo "== [N][K] rhs, K past the cost tables' fit (8192)"
timeout 1500 python tools/ab_algos.py --rounds 5 --algos auto,lp256w4,lp256x192,lp192x192,lp256qm 2320x5904x14336 4096x3584x16384 3840x4096x12288 3072x3072x16384 3584x3584x14336 2560x5120x16384 3000x5000x12288 4000x4000x16384 2048x4096x16384
echo "== row-major rhs"
timeout 1500 python tools/ab_algos.py --nn --rounds 5 --algos auto,lp256w4,lp256x192,lp192x192,lp256qm 2320x5904x14336 4096x3584x16384 3840x4096x12288 3072x3072x16384 3584x3584x14336 2560x5120x16384 3000x5000x12288 4000x4000x16384 2048x4096x16384
} > gpurun_out/r06_long_k_tiles_ab.txt 2>&1
cat gpurun_out/r06_long_k_tiles_ab.txt
