#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2; do
for so in cubecl_amd/csrc/libmi355cube.so cubecl_amd/csrc/variants/*.so; do
  MI355CUBE_LIB=$PWD/$so timeout 120 python tools/dev/split_ab.py 2>&1 | tail -n 1
done
done
