set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_gemm.py -q --no-header -p no:cacheprovider -x -k "lp256qm or auto_selection or output_bound" --timeout 600 2>&1 | tail -15
{
for algo in 15 7 15 7; do timeout 120 python tools/c5_probe.py 6 nn 512 $algo; done
timeout 600 python tools/ab_algos.py --nn --rounds 5 --algos lp256w4,lp256q,lp256qm 8192x8192x8192 8192x8192x2048 8192x8192x4096 8192x4096x2048 8192x8192x1024 8192x8192x512 12288x8192x2048 6144x4096x8192
} > gpurun_out/r06_qm_nn_ab.txt 2>&1
cat gpurun_out/r06_qm_nn_ab.txt
