#!/bin/bash
# GPU box: power-of-two-pitch transposes for the product and every variant library
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2; do
for so in cubecl_amd/csrc/libmi355cube.so cubecl_amd/csrc/variants/*.so; do
  echo "== $(basename $so)"; MI355CUBE_LIB=$PWD/$so timeout 200 python tools/dev/copy_probe.py --transpose 2>&1 | grep -E "2-D transpose \(|8-byte transpose 8192" | cut -c1-40,95-140
done
done
