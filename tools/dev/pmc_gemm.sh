#!/bin/bash
# GPU box: tools/dev/pmc_gemm.sh TAG ALGO "m,n,k" "CTR CTR ..." ["CTR ..."]  -- one rocprofv3 --pmc pass per counter group over tools/dev/gemm_probe.py
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
TAG=$1; ALGO=$2; SPEC=$3; shift 3
i=0
for grp in "$@"; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pg_${TAG}_$i -o p -- python $R/tools/dev/gemm_probe.py $ALGO $SPEC > $O/pg_${TAG}_$i.log 2>&1 )
  f=$(find $O/pg_${TAG}_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" "$TAG" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "gemm_" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("pmc", sys.argv[2], {c: round(sum(v) / len(v)) for c, v in agg.items()}, "n", len(next(iter(agg.values()), [])))
PY
done
