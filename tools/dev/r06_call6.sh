set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python tools/ab_algos.py --rounds 5 --algos lp256w4,lp256m16,lp256q,lp256qm 8192x8192x8192 6144x6144x6144 8192x8192x16384 8192x8192x6144 12288x12288x4096 5120x5120x5120 8192x4096x8192 > gpurun_out/r06_qm_longk_ab.txt 2>&1
cat gpurun_out/r06_qm_longk_ab.txt
