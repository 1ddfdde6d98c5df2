set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python tools/ab_algos.py --rounds 7 --algos lp256w4,lp256m16,lp256qm 4096x4096x4096 4096x4096x8192 4096x4096x2048 3584x3584x3584 4096x4096x1024 > gpurun_out/r06_one_round_m16_ab.txt 2>&1; cat gpurun_out/r06_one_round_m16_ab.txt
for seed in 601 602 603 604; do timeout 1500 python tools/dev/random_audit.py $seed 96; done > gpurun_out/r06_random_audit.txt 2>&1
grep -c "AUTO ->" gpurun_out/r06_random_audit.txt; grep "BEHIND" gpurun_out/r06_random_audit.txt
