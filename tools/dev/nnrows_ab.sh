#!/bin/bash
# Dev (GPU box): few rows x row-major weight -- gemm_nnrows against the 128x128 kernel's native NN form and the [N][K] twins.
cd "$(dirname "$0")/../.."
SHAPES="1x8192x8192 2x8192x8192 4x8192x8192 8x8192x8192 16x8192x8192 1x4096x4096 4x4096x4096 16x4096x4096 1x28672x8192 4x28672x8192 16x28672x8192 4x8192x28672 16x8192x28672 1x14336x4096 4x14336x4096 4x4096x14336 16x4096x14336 1x128256x4096 16x128256x4096 4x32000x4096 16x32000x4096 4x16384x16384 4x65536x2048 16x65536x2048 8x2048x2048 4x1024x8192 4x2048x4096"
echo "== row-major weight [K][N]: auto / lp128 / nnrows (us, cold operands, median of 5 interleaved rounds)"
timeout 400 python tools/ab_algos.py --nn --algos auto,lp128,nnrows $SHAPES
echo "== the same shapes with the weight stored [N][K]: auto"
timeout 300 python tools/ab_algos.py --algos auto $SHAPES
for W in 2; do
  echo "== nnrows with MI355_NNROWS_WG_PER_CU=$W"
  MI355_NNROWS_WG_PER_CU=$W timeout 200 python tools/ab_algos.py --nn --algos nnrows 1x8192x8192 4x8192x8192 16x8192x8192 1x4096x4096 4x28672x8192 1x14336x4096 4x16384x16384
done
for S in 1024 256; do
  echo "== nnrows with MI355_NNROWS_STRIP=$S"
  MI355_NNROWS_STRIP=$S timeout 200 python tools/ab_algos.py --nn --algos nnrows 1x8192x8192 4x8192x8192 1x4096x4096 4x28672x8192 1x14336x4096 4x16384x16384 1x128256x4096
done
echo "== nnrows with MI355_NNROWS_STRIP=512 (above four rows)"
MI355_NNROWS_STRIP=512 timeout 200 python tools/ab_algos.py --nn --algos nnrows 8x8192x8192 16x8192x8192 16x28672x8192 16x4096x14336 16x128256x4096
