#!/bin/bash
# GPU box: the C3 "data movement" levers as interleaved A/Bs of the 8192^3 bf16 GEMM.
#   variants: cubecl_amd/csrc/variants/libmi355cube_gm{2,4,16,32}.so  (XCD patch GROUP_M x 32/GROUP_M; product = 8 x 4)
#   per variant: sustained TFLOP/s (2 interleaved rounds), then one PMC pass each for L2 hits/misses and fabric read bytes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
LIBS="cubecl_amd/csrc/libmi355cube.so $(ls cubecl_amd/csrc/variants/libmi355cube_gm*.so)"
for rep in 1 2 3; do
  for so in $LIBS; do
    echo -n "time rep$rep $(basename $so .so): "
    MI355CUBE_LIB=$R/$so timeout 200 python tools/dev/gemm_probe.py 5 8192,8192,8192 2>&1 | tail -n 1
  done
done
for so in $LIBS; do
  tag=$(basename $so .so)
  for grp in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
    key=$(echo $grp | cut -d' ' -f1)
    ( cd /tmp && MI355CUBE_LIB=$R/$so timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/c3_${tag}_$key -o p -- python $R/tools/dev/gemm_probe.py 5 8192,8192,8192 > $O/c3_${tag}_$key.log 2>&1 )
    f=$(find $O/c3_${tag}_$key -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python - "$f" "$tag" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "gemm_lp256w4" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("pmc", sys.argv[2], {c: round(sum(v) / len(v)) for c, v in agg.items()}, "launches", {c: len(v) for c, v in agg.items()})
PY
  done
done
