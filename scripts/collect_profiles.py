#!/usr/bin/env python3
"""Copy the summaries of gpurun_out/final (scripts/final_profile.sh) into profiles/<round>/.
gpurun merges into gpurun_out/, so older runs' files may lie beside the new ones: the newest of
each kind is taken.  usage: python scripts/collect_profiles.py r2_final"""
import glob, json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(ROOT, "gpurun_out", "final")
P = os.path.join(ROOT, "profiles", sys.argv[1])
os.makedirs(os.path.join(P, "small"), exist_ok=True)

def newest(pattern):
    c = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    assert c, pattern
    return c[-1]

for f in ("bench.json", "bench_20.json", "bench_extra.json", "engines_100k.json", "engines_1m.json", "engines_lidar_raw.json",
          "event_overhead.txt", "sequence_200.json", "sequence_200_noprefetch.json", "sequence_200_map.json", "summary.txt", "ab_r1_r2.json", "threshold_sweep.json",
          "ab_fuse_finish.json", "coarse_clock.json", "ab_small.json", "small_clock.json", "batch_timing.json", "iteration_sensitivity.json",
          "stream_threads.txt", "ab_knn_lists.txt", "normals_timing.json", "ab_nn_bounded.txt", "ab_nn_bounded_shard.txt",
          ):   # (the fuzz logs are trimmed to their last lines by hand)
    if os.path.exists(os.path.join(F, f)) and os.path.getsize(os.path.join(F, f)) > 0:
        shutil.copy(os.path.join(F, f), os.path.join(P, f))
shutil.copy(newest(F + "/stats/**/*kernel_stats.csv"), P + "/kernel_stats.csv")
for g in ("fetch_size", "write_size", "grbm_gui_active", "sq_insts_valu", "sq_waves"):
    shutil.copy(newest(F + "/pmc_%s/**/*counter_collection.csv" % g), P + "/pmc_%s.csv" % g)
shutil.copy(newest(F + "/small/stats/**/*kernel_stats.csv"), P + "/small/kernel_stats.csv")
for f in ("sequence_40.json", "sequence_40_under_rocprof.json"):
    shutil.copy(os.path.join(F, "small", f), os.path.join(P, "small", f))
t = open(os.path.join(F, "shard_overhead.json")).read()
open(os.path.join(P, "shard_overhead.json"), "w").write(t[t.index("{"):])      # (RCCL prints its banner on stdout)
# summary.txt from the files just collected (final_profile.sh's two parts run on different boxes, each box's own
# summary covers its part only)
import subprocess, tempfile
tmp = tempfile.mkdtemp(prefix="icpmi_sum_")
os.makedirs(tmp + "/main"); os.makedirs(tmp + "/small")
shutil.copy(P + "/kernel_stats.csv", tmp + "/main/main_kernel_stats.csv")
shutil.copy(P + "/small/kernel_stats.csv", tmp + "/small/small_kernel_stats.csv")
for g in ("fetch_size", "write_size", "grbm_gui_active", "sq_insts_valu", "sq_waves"):
    os.makedirs(tmp + "/pmc_" + g)
    shutil.copy(P + "/pmc_%s.csv" % g, tmp + "/pmc_%s/x_counter_collection.csv" % g)
s = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "prof_summary.py"), tmp], stdout=subprocess.PIPE, text=True).stdout
open(os.path.join(P, "summary.txt"), "w").write(s.replace(tmp + "/", ""))
shutil.rmtree(tmp, ignore_errors=True)
d = json.load(open(os.path.join(P, "bench.json")))
print("bench30", round(d["value"], 1), round(d["steady_state_it_per_s"], 1), d["roofline"]["avg_launch_ms"], round(d["roofline"]["frac"], 4),
      d["roofline"]["traffic"], round(d["speedup_vs_cpu_1thread"], 1), round(d["cpu_baseline"]["value"], 2))
print(d["stage_ms_untimed_call"]); print(d["parity"]); print("pruned", d["pruned_engine_extra"]["value"], d["resolve_counters"])
print("bench20", json.load(open(os.path.join(P, "bench_20.json")))["value"])
print({k: v["loop_us_per_pass"] for k, v in json.loads(t[t.index("{"):]).items()})
q = json.load(open(os.path.join(P, "sequence_200.json")))
print({k: q[k] for k in ("gpu_ms_per_frame_file_to_pose", "gpu_frame_ms_median", "gpu_frame_ms_p95", "iterations_total", "gated_frames")}, q["oracle"])
q = json.load(open(os.path.join(P, "small", "sequence_40.json")))
print({k: q[k] for k in ("gpu_ms_per_frame_file_to_pose", "gpu_frame_ms_median", "iterations_total")})
print(json.dumps(json.load(open(os.path.join(P, "ab_r1_r2.json")))))
q = json.load(open(os.path.join(P, "sequence_200_map.json")))
print({k: q[k] for k in ("gpu_ms_per_frame_file_to_pose", "gpu_frame_ms_median")}, q["map"]["occupied_cells"], q["oracle"].get("cell_sets_equal"))
print(json.dumps(json.load(open(os.path.join(P, "ab_fuse_finish.json")))["summary_min_over_legs"]))
print(open(os.path.join(P, "coarse_clock.json")).read())
print(json.dumps(json.load(open(os.path.join(P, "bench_extra.json"))))[:1400])
print(json.dumps(json.load(open(os.path.join(P, "ab_small.json")))["summary_min_over_legs"]))
print(open(os.path.join(P, "batch_timing.json")).read())
print(open(os.path.join(P, "small_clock.json")).read())
q = json.load(open(os.path.join(P, "iteration_sensitivity.json")))
print(q["fixtures"], {k: v for k, v in q["stream"].items() if k != "worst"})
print([(r["theta"], "%.2e" % r["kappa"], r["iterations"]) for r in q["kappa_sweep"]])
print(open(os.path.join(P, "stream_threads.txt")).read())
