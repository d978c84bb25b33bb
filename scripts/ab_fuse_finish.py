"""Same-box A/B of the single-GPU loop's fused kernel (k_finish_step_transform, ICPMI_FUSE_FINISH=1,
the default) against the two separate kernels (k_finish_step + k_transform, ICPMI_FUSE_FINISH=0), in
two regimes: the C3 call (100k->100k, 30 iterations) and registrations of filtered LiDAR frames
(50 iterations max, reference defaults).  The knob is read once per process, so every leg is a
child process; the legs alternate A B A B to average out clock drift.  Prints one JSON object."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, json, os
sys.path.insert(0, %r)
import numpy as np, torch
from lidar_slam_from_scratch_amd import capi, synth
out = {}
ctx = capi.Context(device=0)
rng = np.random.default_rng(7)
# C3: uniform random clouds, a small rigid offset, 30 iterations (tolerance 0 so that all of them run)
src, tgt, _ = synth.c3_uniform()
d_src = torch.as_tensor(src, device="cuda"); d_tgt = torch.as_tensor(tgt, device="cuda")
cfg = capi.Context.make_config(max_iterations=30, tolerance=0.0, min_error=0.0)
torch.cuda.synchronize()
ctx.align_device(d_src.data_ptr(), src.shape[0], d_tgt.data_ptr(), tgt.shape[0], cfg)
ts = []
for _ in range(8):
    t0 = time.perf_counter(); res, hist = ctx.align_device(d_src.data_ptr(), src.shape[0], d_tgt.data_ptr(), tgt.shape[0], cfg); ts.append(time.perf_counter() - t0)
out["c3_call_ms"] = round(1e3 * min(ts), 4); out["c3_it_per_s"] = round(30 / min(ts), 1); out["c3_final_error"] = res.final_error
out["c3_hist_tail"] = hist[-1]
A = synth.lidar_frame(3, voxel=0.5, **synth.DRIVE_200)
B = synth.lidar_frame(4, voxel=0.5, **synth.DRIVE_200)
for keep in (500, 2000, 8000):
    a = np.ascontiguousarray(A[np.sort(rng.choice(A.shape[0], min(keep, A.shape[0]), replace=False))])
    b = np.ascontiguousarray(B[np.sort(rng.choice(B.shape[0], min(keep, B.shape[0]), replace=False))])
    cfg = capi.Context.make_config()
    ctx.align(b, a, cfg)
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); res, hist = ctx.align(b, a, cfg); ts.append(time.perf_counter() - t0)
    out["frame_%%d" %% a.shape[0]] = {"align_ms": round(1e3 * min(ts), 4), "iterations": res.num_iterations, "final_error": res.final_error}
print(json.dumps(out))
''' % ROOT
legs = []
for rep in range(3):
    for knob in ("0", "1"):
        env = dict(os.environ, ICPMI_FUSE_FINISH=knob)
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        if r.returncode != 0:
            print(r.stderr[-2000:], file=sys.stderr)
            sys.exit(1)
        legs.append({"fuse_finish": int(knob), **json.loads(r.stdout.strip().splitlines()[-1])})
def best(knob, key, sub=None):
    vals = [(l[key][sub] if sub else l[key]) for l in legs if l["fuse_finish"] == knob]
    return min(vals)
summary = {"c3_call_ms": {"separate": best(0, "c3_call_ms"), "fused": best(1, "c3_call_ms")}}
for k in legs[0]:
    if k.startswith("frame_"):
        summary[k + "_align_ms"] = {"separate": best(0, k, "align_ms"), "fused": best(1, k, "align_ms"),
                                    "iterations": [legs[0][k]["iterations"], legs[1][k]["iterations"]]}
print(json.dumps({"summary_min_over_legs": summary, "legs": legs}, indent=1))
