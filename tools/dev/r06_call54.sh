set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
echo "== 17-64 rows whose 128-column tiles leave a second round mostly empty (N = 33k ... 40k), [N][K] rhs"
for k in 1024 2048 4096 8192; do
  S=""
  for m in 18 24 34 47 61; do for n in 33024 35064 37312 38624 40960; do S="$S ${m}x${n}x${k}"; done; done
  timeout 900 python tools/ab_algos.py --rounds 5 --algos auto,lp128,stream64,lp192x192 $S
done
} > gpurun_out/r06_few_rows_part_round_ab.txt 2>&1
cat gpurun_out/r06_few_rows_part_round_ab.txt
