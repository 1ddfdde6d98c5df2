set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
echo "== one round of the square tile at most, 8-15 K-tiles: narrow tiles in the table from 8 K-tiles (was 16); [N][K] rhs"
S="8840x960x512 4288x2304x512 2368x3584x512 3072x3072x512 4096x4096x512 3584x2048x512 2048x2048x512 1536x6144x512 5120x1280x512 2816x2816x640 3328x4096x640 8840x960x768 4288x2304x768 2368x3584x768 3072x3072x768 4096x4096x768 2048x2048x768 1536x6144x768 2560x2560x896 3840x3072x896 6144x1536x960 2304x2304x960 1792x3584x512 7168x1792x640"
timeout 1500 python tools/ab_algos.py --rounds 5 --algos auto,lp128,lp256x128,lp256w4,lp256x192,lp192x192,lp256qm $S
echo "== row-major rhs"
timeout 1500 python tools/ab_algos.py --nn --rounds 5 --algos auto,lp128,lp256x128,lp256w4,lp256x192,lp192x192,lp256qm $S
echo "== few rows x row-major weight, N not in whole 256-column strips: the strip kernel against the 128 x 128 kernel"
timeout 1500 python tools/ab_algos.py --nn --rounds 5 --algos auto,lp128,nnrows 2x19584x3072 3x34672x1536 4x20472x14336 3x20944x14336 2x14920x8192 4x14920x8192 5x14920x8192 6x14920x8192 8x14920x8192 2x53432x1024 4x53432x1024 5x53432x1024 3x10008x4096 4x30000x4096 2x9000x8192 4x9000x8192 6x9000x8192
echo "== few columns, long K: the streaming kernel against split K"
timeout 1500 python tools/ab_algos.py --rounds 5 --algos auto,lp128,stream64 19152x6x14336 6872x4x14336 9312x4x14336 9312x8x14336 9312x12x14336 9312x16x14336 9312x24x14336 19152x4x12288 19152x8x16384 19152x16x14336 4096x6x14336 32768x6x14336 9312x6x10240 9312x6x24576 36x37624x14336 24x37624x14336 4x8632x14336 8x8632x14336
} > gpurun_out/r06_fresh_seed_fixes_ab.txt 2>&1
cat gpurun_out/r06_fresh_seed_fixes_ab.txt
