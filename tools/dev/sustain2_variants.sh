#!/bin/bash
# usage (GPU box): tools/dev/sustain2_variants.sh "<algos>" shape... -- steady-state TF for the product .so and every variant .so, interleaved twice
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ALGOS=$1; shift
for rep in 1 2; do
for so in cubecl_amd/csrc/libmi355cube.so cubecl_amd/csrc/variants/*.so; do
  echo "== $(basename $so)"; MI355CUBE_LIB=$PWD/$so python tools/dev/sustain2.py "$ALGOS" "$@"
done
done
