set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
echo "== K = 256 ... 448 over many rounds of square tiles"
for nn in "" "--nn"; do
echo "-- $nn"
timeout 900 python tools/ab_algos.py $nn --rounds 5 --algos auto,lp128,lp256w4,lp256p,lp256q,lp256qm,lp256m16 16384x8192x256 11648x12096x256 9856x11072x256 12288x12288x256 16384x16384x256 8192x12288x256 10240x10240x256 11648x12096x320 16384x8192x320 11648x12096x384 16384x16384x384 11648x12096x448
done
echo "== K = 256 ... 448 on at most one round of square tiles"
for nn in "" "--nn"; do
echo "-- $nn"
timeout 900 python tools/ab_algos.py $nn --rounds 5 --algos auto,lp128,lp256x128,lp256w4,lp256x192,lp192x192 1048x6656x256 2048x4096x256 3072x3072x256 4096x4096x256 2368x3584x256 8840x960x256 1536x6144x256 1048x6656x384 3072x3072x384 4096x4096x384 2368x3584x384 8840x960x384 3584x3584x448 2560x4096x320
done
} > gpurun_out/r06_short_k_ab.txt 2>&1
cat gpurun_out/r06_short_k_ab.txt
