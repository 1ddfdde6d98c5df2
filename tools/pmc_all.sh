#!/bin/bash
# Runs on the GPU box (via gpurun): every PMC figure bench.py quotes, taken on the binary that is in the tree NOW, each
# entry stamped with the demangled kernel name rocprofv3 reports, the git sha handed in as $1 (the box has no .git), the
# sha256 of the kernel's sources (bench.kernel_source_sha) and the date.  bench.py refuses an entry whose kernel name or
# source hash does not match what it runs (bench._pmc_entry).
#   passes (separate, MI355X_MICROARCH.md "rocprofv3 PMC slots": FETCH_SIZE and WRITE_SIZE do not fit one pass):
#     1. --pmc FETCH_SIZE          headline GEMM (bench.py --no-extras)     x2 on gfx950 (wide coalesced reads are tallied at 1/2)
#     2. --pmc WRITE_SIZE          headline GEMM                            as reported (uncalibrated)
#     3. --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE   headline GEMM     mfma_util = (busy / 1024 SIMDs) / (active / 8 XCDs)
#     4. --pmc FETCH_SIZE          1 GiB reductions (tools/reduce_probe.py)
#     5. --pmc TCC_HIT_sum TCC_MISS_sum                   headline GEMM     L2 hit rate
#     6-9. the same four on config C5 as benched at N = 1 (tools/c5_probe.py: batch 512 x 2048^3 bf16; gemm_lp256qm.hip since round 6)
# Only --kernel-trace accompanies --pmc (gpurun refuses --pmc with the sys / hip / hsa trace domains).
# Writes gpurun_out/pmc_traffic.json and gpurun_out/pmc_mfma_util.json (copy both to profiles/).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
GIT_SHA=${1:-unknown}; SIZE=${2:-8192}
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
run_pass() {  # name, counters..., -- command
  local name=$1; shift; local ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  ( cd /tmp && timeout 600 rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d $O/pmc_$name -o p -- "$@" > $O/pmc_$name.log 2>&1 )
  echo "pass $name: exit $?"
}
BENCH=(python $R/bench.py --no-extras --no-cpu-baseline --size $SIZE --steps 10 --warmup 3)
run_pass fetch FETCH_SIZE -- "${BENCH[@]}"
run_pass write WRITE_SIZE -- "${BENCH[@]}"
run_pass mfma SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- "${BENCH[@]}"
run_pass l2 TCC_HIT_sum TCC_MISS_sum -- "${BENCH[@]}"
run_pass reduce FETCH_SIZE -- python $R/tools/reduce_probe.py
C5=(python $R/tools/c5_probe.py 4)
run_pass c5_fetch FETCH_SIZE -- "${C5[@]}"
run_pass c5_write WRITE_SIZE -- "${C5[@]}"
run_pass c5_mfma SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- "${C5[@]}"
run_pass c5_l2 TCC_HIT_sum TCC_MISS_sum -- "${C5[@]}"
grep -h "^C5 " $O/pmc_c5_*.log | head -4
python - "$SIZE" "$GIT_SHA" <<'PY'
import collections, csv, datetime, glob, json, statistics, sys
sys.path.insert(0, ".")
import bench
size, git_sha = int(sys.argv[1]), sys.argv[2]
stamp = {"git_sha": git_sha, "date": datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%MZ"), "tool": "tools/pmc_all.sh"}

def rows_of(name):
    f = glob.glob(f"gpurun_out/pmc_{name}/**/*counter_collection.csv", recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []

def med(rows, ctr, kernel):
    v = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == ctr and kernel in r["Kernel_Name"]]
    return (statistics.median(v), len(v)) if v else (None, 0)

def kname(rows, kernel):
    for r in rows:
        if kernel in r["Kernel_Name"]:
            return r["Kernel_Name"]
    return None

traffic, K = {}, bench.HEADLINE_KERNEL
fetch, nf = med(rows_of("fetch"), "FETCH_SIZE", K)
write, nw = med(rows_of("write"), "WRITE_SIZE", K)
if fetch is not None and write is not None:
    fb, wb = int(fetch * 1024 * 2), int(write * 1024)
    traffic[f"gemm_bf16_{size}_algo{bench.HEADLINE_ALGO}"] = dict(stamp, kernel=kname(rows_of("fetch"), K), source_sha=bench.kernel_source_sha("gemm"),
        hbm_bytes_per_launch=fb + wb, fetch_bytes=fb, write_bytes=wb, algorithmic_bytes=3 * size * size * 2,
        FETCH_SIZE_KiB_raw=fetch, WRITE_SIZE_KiB_raw=write, launches=nf,
        note="rocprofv3 --pmc, one counter per pass, medians over the launches of one bench.py --no-extras run; FETCH_SIZE x2 "
             "(gfx950 correction, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported (uncalibrated)")
else:
    print("!! no headline GEMM rows in the FETCH_SIZE / WRITE_SIZE passes")
hit, _ = med(rows_of("l2"), "TCC_HIT_sum", K)
miss, _ = med(rows_of("l2"), "TCC_MISS_sum", K)
if hit is not None and miss is not None and f"gemm_bf16_{size}_algo{bench.HEADLINE_ALGO}" in traffic:
    traffic[f"gemm_bf16_{size}_algo{bench.HEADLINE_ALGO}"].update(TCC_HIT_sum=hit, TCC_MISS_sum=miss, l2_hit_rate=round(hit / (hit + miss), 4))
agg = collections.defaultdict(list)
for r in rows_of("reduce"):
    if "reduce_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if "reduce_kernel<0, 0," in k: name = "sum"                  # <VOP, AOP, DT>: value op SUM = 0, no index op
    elif "reduce_kernel<-1, 1," in k: name = "argmax"            # no value op, index op ARGMAX = 1
    elif "reduce_kernel<0, 1," in k: name = "sum_argmax"
    else: continue
    traffic[f"reduce_1GiB_{name}"] = dict(stamp, kernel=k[:120], source_sha=bench.kernel_source_sha("reduce"),
        fetch_bytes=int(statistics.median(v) * 1024 * 2), algorithmic_bytes=1 << 30, FETCH_SIZE_KiB_raw=statistics.median(v), launches=len(v))
# config C5 as benched (batch 512 x 2048^3 bf16 on the persistent dripped-store kernel, 16x16x32 MFMAs)
KQ = bench.C5_KERNEL
f5, n5 = med(rows_of("c5_fetch"), "FETCH_SIZE", KQ)
w5, _ = med(rows_of("c5_write"), "WRITE_SIZE", KQ)
if f5 is not None and w5 is not None:
    fb, wb = int(f5 * 1024 * 2), int(w5 * 1024)
    ent = dict(stamp, kernel=kname(rows_of("c5_fetch"), KQ), source_sha=bench.kernel_source_sha("gemm_q"), hbm_bytes_per_launch=fb + wb,
               fetch_bytes=fb, write_bytes=wb, algorithmic_bytes=3 * 512 * 2048 * 2048 * 2, FETCH_SIZE_KiB_raw=f5, WRITE_SIZE_KiB_raw=w5, launches=n5,
               note="tools/c5_probe.py under rocprofv3 --pmc, one counter per pass, medians over its launches; FETCH_SIZE x2 (gfx950 correction), WRITE_SIZE as reported")
    h5, _ = med(rows_of("c5_l2"), "TCC_HIT_sum", KQ)
    m5, _ = med(rows_of("c5_l2"), "TCC_MISS_sum", KQ)
    if h5 is not None and m5 is not None:
        ent.update(TCC_HIT_sum=h5, TCC_MISS_sum=m5, l2_hit_rate=round(h5 / (h5 + m5), 4))
    b5, _ = med(rows_of("c5_mfma"), "SQ_VALU_MFMA_BUSY_CYCLES", KQ)
    a5, _ = med(rows_of("c5_mfma"), "GRBM_GUI_ACTIVE", KQ)
    if b5 and a5:
        ent.update(SQ_VALU_MFMA_BUSY_CYCLES=b5, GRBM_GUI_ACTIVE=a5, mfma_util=round((b5 / 1024) / (a5 / 8), 4),
                   expected_busy_cycles_16_per_mfma=16 * 512 * 2048 ** 3 / (16 * 16 * 32))
    traffic["gemm_bf16_c5_batch512"] = ent
else:
    print("!! no C5 rows in the FETCH_SIZE / WRITE_SIZE passes")
json.dump(traffic, open("gpurun_out/pmc_traffic.json", "w"), indent=1)
busy, nb = med(rows_of("mfma"), "SQ_VALU_MFMA_BUSY_CYCLES", K)
active, _ = med(rows_of("mfma"), "GRBM_GUI_ACTIVE", K)
util = {}
if busy and active:
    util[f"gemm_bf16_{size}"] = dict(stamp, kernel=kname(rows_of("mfma"), K), source_sha=bench.kernel_source_sha("gemm"),
        SQ_VALU_MFMA_BUSY_CYCLES=busy, GRBM_GUI_ACTIVE=active, mfma_busy_cycles_per_simd=busy / 1024, resident_cycles_per_xcd=active / 8,
        mfma_util=round((busy / 1024) / (active / 8), 4), expected_busy_cycles_16_per_mfma=16 * size ** 3 / (16 * 16 * 32), launches=nb,
        note="rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace, medians over the launches of one bench.py --no-extras run")
else:
    print("!! no headline GEMM rows in the MFMA pass")
json.dump(util, open("gpurun_out/pmc_mfma_util.json", "w"), indent=1)
print(json.dumps({"traffic": traffic, "mfma_util": util})[:3000])
PY
