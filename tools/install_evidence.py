#!/usr/bin/env python3
"""Copies what one evidence call (tools/pmc_all.sh SHA; tools/gpu_check.sh TAG; tools/gpu_rehearse_n2.sh) left in gpurun_out/ into
profiles/ under the round's names, and splits the headline rocprof trace by phase.  usage: tools/install_evidence.py TAG SHA"""
import csv, glob, json, shutil, sys
tag, sha = sys.argv[1], sys.argv[2]
G, P = "gpurun_out/", "profiles/"
shutil.copy(G + f"{tag}_bench.json", P + f"{tag}_bench.json")
shutil.copy(G + f"{tag}_pytest.log", P + f"{tag}_pytest_gpu.log")
shutil.copy(G + f"{tag}_prof/bench_kernel_stats.csv", P + f"{tag}_rocprof_kernel_stats_headline.csv")
shutil.copy(G + f"{tag}_prof_all/bench_kernel_stats.csv", P + f"{tag}_rocprof_kernel_stats_all_configs.csv")
shutil.copy(glob.glob(G + f"{tag}_prof_reduce/**/*kernel_stats.csv", recursive=True)[0], P + f"{tag}_rocprof_kernel_stats_reduce_1GiB.csv")
line = [l for l in open(G + f"{tag}_prof.log") if l.startswith("{")][-1]
open(P + f"{tag}_bench_under_rocprof_headline.json", "w").write(line)
under = json.loads(line)
for src, dst in (("rehearse_n2.json", f"{tag}_rehearse_n2.json"), ("select_audit.txt", f"{tag}_select_audit.txt"), ("select_audit_nn.txt", f"{tag}_select_audit_nn.txt"),
                 (f"{tag}_kres.txt", f"{tag}_kernel_resources.txt"), ("parity_margins.jsonl", f"{tag}_parity_margins.jsonl"),
                 ("pmc_traffic.json", "pmc_traffic.json"), ("pmc_mfma_util.json", "pmc_mfma_util.json")):
    shutil.copy(G + src, P + dst)
sys.path.insert(0, ".")
import bench
HK = bench.HEADLINE_KERNEL
rows = [r for r in csv.DictReader(open(glob.glob(G + f"{tag}_prof/**/*kernel_trace.csv", recursive=True)[0])) if HK.split("<")[0] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
n, t, w = len(d), d[-30:], d[:-30]
text = f"""rocprofv3 --kernel-trace --stats -- python bench.py --no-extras --no-cpu-baseline (tools/gpu_check.sh {tag}), tree {sha}
kernel {HK}: {n} launches in the trace, in start order:
  5 warm-up + {len(w) - 5} plateau warm-up launches     average {sum(w) / len(w):8.1f} us   (the DVFS ramp from idle: first 20 average {sum(d[:20]) / 20:.1f})
  the 30 TIMED launches (back to back)            average {sum(t) / 30:8.1f} us   min {min(t):.1f}  max {max(t):.1f}
  all {n}                                          average {sum(d) / n:8.1f} us   (what {tag}_rocprof_kernel_stats_headline.csv prints)
bench.py's figure for the timed region of the same process (HIP events around the 30 launches): roofline.kernel_ms = {under["roofline"]["kernel_ms"]} ms
({tag}_bench_under_rocprof_headline.json); kernel_ms x shader clock there = {under["roofline"].get("kernel_ms_times_shader_clock_GHz")} (GHz x ms: the box-independent figure).
The per-sample launches behind frac_per_sample_median run with the extras only.
"""
open(P + f"{tag}_rocprof_headline_timed_region.txt", "w").write(text)
print(text)
r = json.loads([l for l in open(P + f"{tag}_bench.json") if l.startswith("{")][-1])
rf = r["roofline"]
print({k: r[k] for k in ("value", "ms_per_step")}, {k: rf[k] for k in ("frac", "kernel_ms", "traffic", "mfma_util_pmc", "shader_clock_GHz", "frac_of_peak_at_clock", "frac_per_sample_median",
      "reduce_sum_achieved_GBs", "reduce_sum_frac", "reduce_sum_frac_per_sample_median", "reduce_argmax_frac", "reduce_sum_argmax_fused_frac", "c5_batched_512x2048_achieved_TFLOPs_whole_job",
      "c5_batched_512x2048_frac", "c5_batched_512x2048_mfma_util_pmc", "c5_batched_512x2048_l2_hit_rate", "c2_f32_4096_NT_achieved_TFLOPs", "c2_f32_4096_NT_frac", "c2_f32_4096_NT_frac_per_sample_median")})
print(r["cpu_baseline"]["value"], r["cpu_baseline"]["reduce_value"], json.dumps(r["extra"]["measured_ceilings"]))
g = r["extra"]["gemm_bf16_shapes"]
print({k: (v["back_to_back_ms"], v["algo"]) for k, v in g.items() if "8192x8192" in k or "_NN" in k})
