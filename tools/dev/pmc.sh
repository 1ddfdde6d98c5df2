#!/bin/bash
# usage (GPU box): tools/dev/pmc.sh TAG "<bench args>" "CTR1 CTR2" ["CTR3 ..." ...]  -- one rocprofv3 --pmc pass per counter group
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=$1; ARGS=$2; shift 2
export TMPDIR=/tmp
R=$PWD
i=0
for grp in "$@"; do
  i=$((i+1))
  ( cd /tmp && timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_$i -o p -- python $R/bench.py $ARGS > $R/gpurun_out/pmc_${TAG}_$i.log 2>&1 )
  f=$(find gpurun_out/pmc_${TAG}_$i -name "*counter_collection.csv" | head -1)
  echo "== group $i: $grp -> $f"
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "fill" in k or "memset" in k.lower(): continue
    print(k, {c: (round(sum(v)/len(v), 1), len(v)) for c, v in d.items()})
PY
done
