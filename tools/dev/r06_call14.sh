set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
timeout 300 python tools/dev/sync_gap_probe.py 8192
echo "== warm vs cold operands (is the fabric fetch served by the Infinity Cache?)"
echo "-- cold (operand sets rotate through > 768 MiB)"; timeout 600 python tools/ab_algos.py --rounds 5 --algos lp256qm,lp256m16 8192x8192x8192 8192x8192x4096 8192x8192x2048
echo "-- warm (one operand set)"; timeout 600 python tools/ab_algos.py --warm --rounds 5 --algos lp256qm,lp256m16 8192x8192x8192 8192x8192x4096 8192x8192x2048
} > gpurun_out/r06_c3_sync_and_mall.txt 2>&1
cat gpurun_out/r06_c3_sync_and_mall.txt
cd /tmp && export TMPDIR=/tmp && (rocprofv3 --list-avail 2>/dev/null | grep -i -E "mall|hbm|dram|EA_RD|EA_WR|MC_|UMC" | head -40) > "$OLDPWD/gpurun_out/r06_counters_avail.txt" 2>&1; cd "$OLDPWD"; wc -l gpurun_out/r06_counters_avail.txt; head -40 gpurun_out/r06_counters_avail.txt
