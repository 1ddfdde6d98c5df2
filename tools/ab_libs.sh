#!/bin/bash
# Runs on the GPU box: the same probe under two (or more) builds of the library, interleaved, several rounds -- the only way to see
# a few-percent change between builds on this pool (box-to-box spread is larger).  usage: tools/ab_libs.sh "libA.so libB.so" ROUNDS probe args...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
LIBS=$1; ROUNDS=$2; shift 2
for rep in $(seq 1 $ROUNDS); do for so in $LIBS; do
  echo "== round $rep $(basename $so)"; MI355CUBE_LIB=$PWD/cubecl_amd/csrc/$so timeout 600 "$@" 2>&1 | grep -v "amdgpu.ids"
done; done
