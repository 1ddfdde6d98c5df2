set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
timeout 700 python tools/ab_algos.py --nn --rounds 5 --algos lp256w4,lp256q,lp256qm 8192x8192x8192 8192x8192x2048 8192x8192x4096 8192x4096x2048 8192x8192x1024 8192x8192x512 12288x8192x2048 6144x4096x8192 8192x8192x16384 5120x5120x5120
timeout 700 python tools/ab_algos.py --rounds 5 --algos lp256w4,lp256m16,lp256q,lp256qm 8192x8192x8192 8192x8192x2048 8192x8192x1024 8192x8192x512 8192x8192x448 6144x6144x6144 4096x4096x4096 5120x5120x2048
} > gpurun_out/r06_qm_pre_shapes_ab.txt 2>&1
cat gpurun_out/r06_qm_pre_shapes_ab.txt
