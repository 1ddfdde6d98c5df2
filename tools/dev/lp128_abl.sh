#!/bin/bash
# GPU box: product and every variant library, interleaved twice
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2; do
for so in cubecl_amd/csrc/libmi355cube.so cubecl_amd/csrc/variants/*.so; do
  case $so in *ks*|*nt*|*pf2.so|*swcvt*|*prio*|*noks*|*pfw*|*libmi355cube.so) chk=1;; *) chk=0;; esac
  CHECK=$chk MI355CUBE_LIB=$PWD/$so timeout 120 python tools/dev/lp128_abl.py 2>&1 | tail -n 1
done
done
