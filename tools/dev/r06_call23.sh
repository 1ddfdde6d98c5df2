cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
for lay in nn nt; do
( cd /tmp && timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmcq_$lay -o p -- python $R/tools/c5_probe.py 3 $lay 512 15 > $O/pmcq_$lay.log 2>&1 )
python - $lay <<'PY'
import csv, glob, statistics, sys, collections
lay=sys.argv[1]
acc=collections.defaultdict(list)
for f in glob.glob(f"gpurun_out/pmcq_{lay}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "lp256qm" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(lay, {k: statistics.median(v) for k,v in acc.items()})
PY
done
