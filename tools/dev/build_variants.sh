#!/bin/bash
# Builds side copies of libmi355cube.so with development switches for A/B and ablation runs:
#   tools/dev/build_variants.sh NAME "-DLP256_ABL=1" [file.hip ...]   -> cubecl_amd/csrc/variants/libmi355cube_NAME.so
set -e
cd "$(dirname "$0")/../../cubecl_amd/csrc"
NAME=$1; FLAGS=$2; shift 2
FILES=${@:-gemm_lp256w4.hip}
mkdir -p variants/obj_$NAME
OBJS=""
for f in $(sed -n 's/^SRCS := //p' Makefile); do
  base=${f%.*}
  if echo " $FILES " | grep -q " $f "; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-value -Wno-unused-result --offload-arch=gfx950 -I../../include $FLAGS -x hip -c $f -o variants/obj_$NAME/$base.o
    OBJS="$OBJS variants/obj_$NAME/$base.o"
  else
    OBJS="$OBJS build/$base.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libmi355cube_$NAME.so $OBJS -ldl -lpthread
echo built variants/libmi355cube_$NAME.so
