#!/bin/bash
# Dev: a fixed set of rocprofv3 --pmc passes over one command, aggregated per kernel (median over its launches) into one table.
#   usage (GPU box): tools/dev/pmc_passes.sh TAG KERNEL_SUBSTRING -- command ...
# Passes (8 SQ slots, 4 TCC slots, 2 GRBM slots per pass; only --kernel-trace accompanies --pmc):
#   sq1 wave / wait / active cycles   sq2 instruction counts + LDS conflicts + MFMA busy   sq3 LDS stalls + GRBM   ta TA / TCP stalls   l2 hit / miss   fetch
# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md); FETCH_SIZE is doubled on gfx950.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=$1; KSUB=$2; shift 2; [ "$1" = "--" ] && shift
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
run_pass() { local name=$1; shift; local ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  ( cd /tmp && timeout 600 rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d $O/pmcp_${TAG}_$name -o p -- "$@" > $O/pmcp_${TAG}_$name.log 2>&1 ); echo "pass $name: exit $?"; }
run_pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -- "$@"
run_pass sq2 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -- "$@"
run_pass sq3 SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE -- "$@"
run_pass ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum -- "$@"
run_pass l2 TCC_HIT_sum TCC_MISS_sum -- "$@"
run_pass fetch FETCH_SIZE -- "$@"
python - "$TAG" "$KSUB" <<'PY'
import collections, csv, glob, statistics, sys
tag, ksub = sys.argv[1], sys.argv[2]
table = collections.defaultdict(dict)     # kernel -> counter -> median
counts = {}
for d in sorted(glob.glob(f"gpurun_out/pmcp_{tag}_*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if ksub in r["Kernel_Name"]:
                acc[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in acc.items():
            table[k][c] = statistics.median(v); counts[k] = len(v)
dur = collections.defaultdict(list)
for f in glob.glob(f"gpurun_out/pmcp_{tag}_sq1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if ksub in r["Kernel_Name"]:
            dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = open(f"gpurun_out/pmcp_{tag}.txt", "w")
for k in sorted(table):
    print(f"== {k[:150]}   launches {counts[k]}   median duration under the sq1 pass {statistics.median(dur[k]) if dur[k] else float('nan'):.1f} us", file=out)
    for c in sorted(table[k]):
        print(f"   {c:40s} {table[k][c]:16.1f}", file=out)
    t = table[k]
    if "SQ_WAVE_CYCLES" in t:
        w = t["SQ_WAVE_CYCLES"]
        print(f"   -- shares of SQ_WAVE_CYCLES: WAIT_ANY {t.get('SQ_WAIT_ANY', 0) / w:.3f}  WAIT_INST_ANY {t.get('SQ_WAIT_INST_ANY', 0) / w:.3f} (of which LDS {t.get('SQ_WAIT_INST_LDS', 0) / w:.3f})  "
              f"ACTIVE_INST_ANY {t.get('SQ_ACTIVE_INST_ANY', 0) / w:.3f} (LDS {t.get('SQ_ACTIVE_INST_LDS', 0) / w:.3f}, VMEM {t.get('SQ_ACTIVE_INST_VMEM', 0) / w:.3f}, VALU {t.get('SQ_ACTIVE_INST_VALU', 0) / w:.3f})", file=out)
    if "SQ_LDS_IDX_ACTIVE" in t and t["SQ_LDS_IDX_ACTIVE"]:
        print(f"   -- LDS bank conflict cycles / LDS active cycles: {t.get('SQ_LDS_BANK_CONFLICT', 0) / t['SQ_LDS_IDX_ACTIVE']:.4f}", file=out)
    if "TCC_HIT_sum" in t:
        print(f"   -- L2 hit rate {t['TCC_HIT_sum'] / (t['TCC_HIT_sum'] + t.get('TCC_MISS_sum', 0)):.4f}", file=out)
    if "FETCH_SIZE" in t:
        print(f"   -- fabric fetch {t['FETCH_SIZE'] * 1024 * 2 / 1e6:.1f} MB per launch (FETCH_SIZE x 2)", file=out)
    if "TCP_TCC_READ_REQ_sum" in t and t["TCP_TCC_READ_REQ_sum"]:
        print(f"   -- mean L1 -> L2 read latency {t.get('TCP_TCC_READ_REQ_LATENCY_sum', 0) / t['TCP_TCC_READ_REQ_sum']:.0f} cycles", file=out)
out.close()
print(open(f"gpurun_out/pmcp_{tag}.txt").read())
PY
