set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
echo "== [N][K] rhs"
timeout 1500 python tools/ab_algos.py --rounds 5 --algos auto,lp256w4,lp256x192,lp192x192,lp256qm 4352x4096x4096 5120x4096x4096 4608x4096x4096 6144x6144x6144 6144x6144x4096 6144x6144x2048 6144x6144x1024 4352x4096x2048 8448x8192x4096 6400x6144x6144 7168x5120x8192 4096x4352x8192 4608x4096x2048 4352x4096x8192
echo "== row-major rhs"
timeout 1500 python tools/ab_algos.py --nn --rounds 5 --algos auto,lp256w4,lp256x192,lp192x192,lp256qm 4352x4096x4096 6144x6144x6144 6144x6144x2048 4608x4096x8192 7168x5120x8192 5120x4096x4096
} > gpurun_out/r06_tail_split_rule_ab.txt 2>&1
cat gpurun_out/r06_tail_split_rule_ab.txt
