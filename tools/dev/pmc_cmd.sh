#!/bin/bash
# usage (GPU box): tools/dev/pmc_cmd.sh TAG "<python script + args>" KERNEL_SUBSTR "CTR1 CTR2" ["CTR3 ..." ...]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=$1; CMD=$2; KSUB=$3; shift 3
export TMPDIR=/tmp; R=$PWD; i=0
for grp in "$@"; do
  i=$((i+1))
  ( cd /tmp && timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_$i -o p -- python $R/$CMD > $R/gpurun_out/pmc_${TAG}_$i.log 2>&1 )
  f=$(find gpurun_out/pmc_${TAG}_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" "$KSUB" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        agg[r["Kernel_Name"][20:75]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v) / 1e6, 3) for c, v in d.items()}, "(x1e6)")
PY
done
