for rep in 1 2; do
for so in cubecl_amd/csrc/libmi355cube.so cubecl_amd/csrc/variants/libmi355cube_oldmap.so; do
  echo "== $so"; MI355CUBE_LIB=$PWD/$so timeout 300 python tools/dev/c5_probe.py 2>&1 | grep "_w4\|_auto" | tr '\n' ' '; echo
done; done
