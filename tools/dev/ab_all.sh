# usage (GPU box): tools/dev/ab_all.sh -- bf16 / fp8 / MX shapes on the product .so and every variant .so
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2; do
for so in cubecl_amd/csrc/libmi355cube.so cubecl_amd/csrc/variants/*.so; do
  echo "== $(basename $so)"
  MI355CUBE_LIB=$PWD/$so timeout 300 python tools/dev/ab_shapes.py 2>&1 | tail -1
done; done
