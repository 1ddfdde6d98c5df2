#!/usr/bin/env python3
"""Writes README.md and profiles/README.md from tools/templates/*.in: the placeholders (C3_TF, C4_GBS, ... EVIDENCE_SHA) become the figures of
profiles/TAG_bench.json and the files next to it, i.e. of the evidence call tools/install_evidence.py TAG SHA installed.  Edit the templates.
usage: tools/fill_evidence_numbers.py TAG SHA"""
import json, re, sys
sys.path.insert(0, ".")
tag, sha = sys.argv[1], sys.argv[2]
P = "profiles/"
r = json.loads([l for l in open(P + f"{tag}_bench.json") if l.startswith("{")][-1])
rf, ex = r["roofline"], r["extra"]
timed = open(P + f"{tag}_rocprof_headline_timed_region.txt").read()
pyt = [l for l in open(P + f"{tag}_pytest_gpu.log") if " passed" in l][-1]
m = re.search(r"(\d+) passed(?:, (\d+) skipped)?", pyt)
fmt = lambda x, n=0: f"{x:,.{n}f}".replace(",", " ")
sub = {
    "EVIDENCE_SHA": sha,
    "C3_TF": fmt(r["value"], 1), "C3_FRAC": f"{rf['frac']:.4f}", "C3_MS": f"{rf['kernel_ms']}", "C3_GHZ": f"{rf['shader_clock_GHz']}",
    "C3_ATCLK": f"{rf['frac_of_peak_at_clock']:.3f}",
    "C3_TRAFFIC": (f"{rf['traffic'] / rf['algorithmic_bytes_per_launch']:.2f}" if rf.get("traffic") else "n/a"),
    "C3_L2": str(json.load(open(P + "pmc_traffic.json")).get(f"gemm_bf16_8192_algo{__import__('bench').HEADLINE_ALGO}", {}).get("l2_hit_rate", "n/a")), "C3_UTIL": str(rf.get("mfma_util_pmc", "n/a")),
    "C4_GBS": fmt(rf["reduce_sum_achieved_GBs"]), "C4_FRAC": f"{rf['reduce_sum_frac']:.3f}", "C4_PS": f"{rf['reduce_sum_frac_per_sample_median']:.3f}",
    "READ_PROBE": fmt(ex["measured_ceilings"]["hbm_read_GBs"]),
    "SHARD_SUM_FRAC": f"{rf['reduce_shard_of_8_sum_frac']:.2f}", "SHARD_FUSED_FRAC": f"{rf['reduce_shard_of_8_fused_frac']:.2f}",
    "SHARD_SUM": f"{rf['reduce_shard_of_8_sum_us']:.1f}", "SHARD_FUSED": f"{rf['reduce_shard_of_8_fused_us']:.1f}",
    "EXCH2": f"{rf['reduce_exchange_one_rank_two_collectives_us']:.1f}", "EXCH": f"{rf['reduce_exchange_one_rank_gather_us']:.1f}",
    "PASSX": f"{rf['reduce_shard_pass_plus_exchange_us']:.1f}", "PROJ": f"{rf['projected_c4_8gpu_us']:.1f}",
    "C5_TF": fmt(rf["c5_batched_512x2048_achieved_TFLOPs_whole_job"]), "C5_FRAC": f"{rf['c5_batched_512x2048_frac']:.3f}",
    "C2_TF": f"{rf['c2_f32_4096_NT_achieved_TFLOPs']}", "C2_FRAC": f"{rf['c2_f32_4096_NT_frac']:.3f}",
    "CPU_TF": f"{r['cpu_baseline']['value']:.3f}", "CPU_GBS": f"{r['cpu_baseline']['reduce_value']:.1f}",
    "CEIL_ONES": fmt(ex["measured_ceilings"]["mfma_bf16_TFLOPs"]), "CEIL_UNI": fmt(ex["measured_ceilings"]["mfma_bf16_uniform_operands_TFLOPs"]),
    "ROCPROF_US": re.search(r"30 TIMED launches.*?average\s+([\d.]+) us", timed).group(1),
    "PYTEST_LINE": f"{m.group(1)} passed, {m.group(2) or 0} skipped",
    "C1_US": f"{ex['sum_things_1M_f32']['back_to_back_us']}",
}
for src, path in (("tools/templates/README.md.in", "README.md"), ("tools/templates/profiles_README.md.in", P + "README.md")):
    s = open(src).read()
    for k in sorted(sub, key=len, reverse=True):          # longest first: SHARD_SUM_FRAC before SHARD_SUM
        s = re.sub(rf"\b{k}\b", sub[k], s)
    open(path, "w").write(s)
    left = sorted(set(re.findall(r"\b[A-Z][A-Z0-9]+_[A-Z0-9_]+\b", s)) & set(sub))
    print(path, "filled;", "left:", left)
