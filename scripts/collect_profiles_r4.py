#!/usr/bin/env python3
"""Copy the summaries of gpurun_out/final4 (scripts/final_profile_r4.sh) into profiles/r4_final/ and print the numbers the
documents quote.  gpurun merges into gpurun_out/, so older runs' files may lie beside the new ones: the newest of each
kind is taken.  usage: python scripts/collect_profiles_r4.py"""
import glob, json, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(ROOT, "gpurun_out", "final4")
P = os.path.join(ROOT, "profiles", "r4_final")
for d in ("", "small", "c2_20k"):
    os.makedirs(os.path.join(P, d), exist_ok=True)

def newest(pattern):
    c = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    assert c, pattern
    return c[-1]

def strip_banner(src, dst):   # (RCCL prints its banner on stdout)
    t = open(src).read()
    open(dst, "w").write(t[t.index("{"):])

for f in ("bench.json", "bench_20.json", "bench_extra.json", "groups_clock.json", "c3_default_timeline.txt", "engine_threshold.json",
          "offset_timing.json", "engines_1m.json", "engines_4m.json", "sequence_200.json", "sequence_200_map.json", "ab_small.json", "batch_timing.json", "coarse_clock.json"):
    p = os.path.join(F, f)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(P, f))
for f in ("shard_overhead_default.json", "shard_overhead_all_pairs.json"):
    if os.path.exists(os.path.join(F, f)):
        strip_banner(os.path.join(F, f), os.path.join(P, f))
for f in ("fuzz_stopping.txt", "fuzz_engines.txt", "fuzz_bounded.txt"):   # the last lines + every MISMATCH / SENSITIVE / UNDER line
    p = os.path.join(F, f)
    if os.path.exists(p):
        lines = open(p).read().splitlines()
        keep = [l for l in lines if l.startswith(("MISMATCH", "SENSITIVE", "UNDER"))] + lines[-3:]
        open(os.path.join(P, f), "w").write("\n".join(keep) + "\n")
shutil.copy(newest(F + "/stats/**/*kernel_stats.csv"), P + "/kernel_stats.csv")
shutil.copy(newest(F + "/c3_default/stats/**/*kernel_stats.csv"), P + "/kernel_stats_default_engine_20_steps.csv")
shutil.copy(newest(F + "/c2_20k/stats/**/*kernel_stats.csv"), P + "/c2_20k/kernel_stats.csv")
shutil.copy(os.path.join(F, "c2_20k", "timeline.txt"), P + "/c2_20k/timeline.txt")
shutil.copy(newest(F + "/small/stats/**/*kernel_stats.csv"), P + "/small/kernel_stats.csv")
shutil.copy(os.path.join(F, "small", "sequence_40_under_rocprof.json"), P + "/small/sequence_40_under_rocprof.json")
groups = ("fetch_size", "write_size", "grbm_gui_active", "sq_insts_valu", "sq_waves")
for eng, name in ((0, "default"), (2, "all_pairs")):
    for g in groups:
        shutil.copy(newest(F + "/pmc%d_%s/**/*counter_collection.csv" % (eng, g)), P + "/pmc_%s_%s.csv" % (name, g))
# summary.txt from the files just collected
tmp = tempfile.mkdtemp(prefix="icpmi_sum_")
def put(sub, src, name):
    os.makedirs(os.path.join(tmp, sub), exist_ok=True)
    shutil.copy(src, os.path.join(tmp, sub, name))
put("bench_both_engines", P + "/kernel_stats.csv", "x_kernel_stats.csv")
put("default_engine_20_steps", P + "/kernel_stats_default_engine_20_steps.csv", "x_kernel_stats.csv")
put("c2_20k", P + "/c2_20k/kernel_stats.csv", "x_kernel_stats.csv")
put("small", P + "/small/kernel_stats.csv", "x_kernel_stats.csv")
for eng, name in ((0, "default"), (2, "all_pairs")):
    for g in groups:
        put("pmc_%s_%s" % (name, g), P + "/pmc_%s_%s.csv" % (name, g), "x_counter_collection.csv")
s = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "prof_summary.py"), tmp], stdout=subprocess.PIPE, text=True).stdout
open(os.path.join(P, "summary.txt"), "w").write(s.replace(tmp + "/", ""))
shutil.rmtree(tmp, ignore_errors=True)
for name in ("bench.json", "bench_20.json"):
    d = json.load(open(os.path.join(P, name)))
    de = d["default_engine"]
    print(name, "value", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 4), "call", {k: round(v, 3) for k, v in d["call_ms"].items() if k != "note"},
          "steady", round(d["steady_state_it_per_s"], 1), "frac", round(d["roofline"]["frac"], 4), "k_ms", round(d["roofline"]["avg_launch_ms"], 4),
          "traffic", d["roofline"].get("traffic"))
    print("   default", round(de["value"], 1), "ms/step", round(de["ms_per_step"], 4), "steady", round(de["steady_state_it_per_s"], 1),
          "coarse ms", round(de["roofline"]["avg_launch_ms"], 4), "culled", round(de["roofline"]["pairs_culled_frac"], 4), "frac", round(de["roofline"]["frac"], 4),
          "bit-equal", de["history_bit_equal"])
    if "cpu_baseline" in d:
        print("   cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["samples_it_per_s"], "pinned", d["cpu_baseline"]["pinned_cpu"], "x", round(d["speedup_vs_cpu_1thread"], 1), d["parity"])
        print("  ", d["stage_ms_untimed_call"], d["resolve_counters"])
print(json.dumps(json.load(open(os.path.join(P, "bench_extra.json"))))[:1800])
for f in ("shard_overhead_default.json", "shard_overhead_all_pairs.json"):
    print(f, {k: v["loop_us_per_pass"] for k, v in json.load(open(os.path.join(P, f))).items()})
q = json.load(open(os.path.join(P, "sequence_200.json")))
print({k: q[k] for k in ("gpu_ms_per_frame_file_to_pose", "gpu_frame_ms_median", "gpu_frame_ms_p95", "iterations_total", "gated_frames")}, q["oracle"])
print(json.dumps(json.load(open(os.path.join(P, "ab_small.json")))["summary_min_over_legs"]))
print(open(os.path.join(P, "groups_clock.json")).read()[:900])
