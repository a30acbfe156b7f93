"""Copies what `scripts/round_end.sh <name>` left under gpurun_out/<name>/ into profiles/<name>/ (the summaries that are kept; traces and
raw counter files stay behind) and the chained-launch evidence into profiles/<name>/chained/.
    python3 scripts/collect_profiles.py r03n_round3_end"""
import glob, os, shutil, sys
name = sys.argv[1]
S, D = f"gpurun_out/{name}", f"profiles/{name}"
K = D + "/chained"   # (round 3 kept its chained-launch evidence in profiles/r03k_chained_launches)
os.makedirs(D, exist_ok=True); os.makedirs(K, exist_ok=True)
def lastline(f):
    return [l for l in open(f) if l.startswith("{")][-1]
shutil.copy(S + "/trace/t_kernel_stats.csv", D + "/kernel_stats.csv")
shutil.copy(S + "/trace_chained/t_kernel_stats.csv", D + "/kernel_stats_chained_default_command.csv")
open(D + "/bench_line_of_the_traced_run.json", "w").write(lastline(S + "/bench_trace.log"))
open(D + "/bench_line_of_the_chained_traced_run.json", "w").write(lastline(S + "/bench_trace_chained.log"))
for f in ["bench_default.json", "bench_default_no_chain.json", "bench_other_workloads.json", "call_sizes.txt", "ragged_call_sizes.txt",
          "multichannel_reverb.txt", "send_filters.txt", "send_filters_with_the_pre_pass_kernel.txt", "update_storm.txt",
          "update_storm_in_stream_order.txt", "update_storm_without_the_cross_fading_build.txt", "per_effect_type.txt", "kinds_presets.txt",
          "low_rates.txt", "host_io.txt", "config4_parts.txt", "chorus_delays.txt", "timeline_steady_kernel.txt", "chain_step.txt",
          "workgroup_placement.txt", "overlap_probe.txt", "ab_round1_vs_round2.txt", "ab_round1_vs_round2_presets.txt",
          "ab_round1_vs_round2_ring_light_types.txt"]:
    if os.path.exists(S + "/" + f):
        shutil.copy(S + "/" + f, D + "/" + f)
for n in ("pmc_sq1", "pmc_sq2", "pmc_fetch", "pmc_write"):
    fs = glob.glob(S + "/" + n + "/**/*summary.txt", recursive=True)
    if fs:
        shutil.copy(fs[0], D + "/" + n + "_summary.txt")
if os.path.exists(S + "/pmc_configs/valu_issue.json"):
    shutil.copy(S + "/pmc_configs/valu_issue.json", D + "/valu_issue.json")
for f in ["acquire_cost.txt", "bench20_spin_up.txt", "bench_steps20_warmup5.txt", "chain_depth.txt", "instances_and_call_sizes.txt",
          "kinds_presets.txt", "send_filters.txt", "storm4_chained_timeline.txt", "timeline_chained.txt", "timeline_stream_order.txt",
          "uncached_memory.txt"]:
    if os.path.exists(S + "/chained/" + f):
        shutil.copy(S + "/chained/" + f, K + "/" + f)
print("copied into", D, "and", K)
