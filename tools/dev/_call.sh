set -x
cd /root/repo
SH=""
for mn in 3072x3072 2168x7344 3648x8832 3200x2496 13536x536 4096x4096 2048x2048 2560x2560 4096x2048 6144x4096 1792x6112 4352x2624 58376x192 5120x5120 1536x1536 12288x1024 3584x3584 2304x2304; do
  for k in 128 256 512; do SH="$SH ${mn}x${k}"; done
done
for mn in 3072x3072 2168x7344 4096x4096 6144x4096 2560x2560; do SH="$SH ${mn}x384 ${mn}x768"; done
timeout 900 python tools/ab_algos.py --rounds 5 --algos auto,lp128,lp256x128,lp256w4,lp256x192,lp192x192,lp256m16 $SH > gpurun_out/r05_tile_short_k_ab.txt 2>&1
timeout 900 python tools/ab_algos.py --nn --rounds 5 --algos auto,lp128,lp256x128,lp256w4,lp256x192,lp192x192 $SH > gpurun_out/r05_tile_short_k_nn_ab.txt 2>&1
tail -5 gpurun_out/r05_tile_short_k_ab.txt gpurun_out/r05_tile_short_k_nn_ab.txt
