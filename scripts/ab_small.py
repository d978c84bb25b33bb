"""Same-box A/B of the small-cloud kernel (icp_small.h, ICPMI_SMALL=1, the default) against the general
path (ICPMI_SMALL=0: k_nn_coarse + k_nn_resolve4 + k_finish_step_transform) on what the reference's
callers register: filtered LiDAR frames.  Per leg: the iteration PERIOD at ~8k -> 8k points (two
device-resident calls of 20 and 60 forced iterations; (t60 - t20) / 40), whole registrations at 500 /
2,000 / 8,000 points with the reference's defaults, and a 60-frame file -> pose stream.  The knob is
read once per process, so every leg is a child process; legs alternate.  Prints one JSON object."""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, json, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "scripts"))
import numpy as np, torch
from lidar_slam_from_scratch_amd import capi, synth, odometry
out = {}
ctx = capi.Context(device=0)
rng = np.random.default_rng(7)
A = synth.lidar_frame(3, voxel=0.4, **synth.DRIVE_200)
B = synth.lidar_frame(4, voxel=0.4, **synth.DRIVE_200)
def sub(P, keep):
    return np.ascontiguousarray(P[np.sort(rng.choice(P.shape[0], min(keep, P.shape[0]), replace=False))])
a, b = sub(A, 8000), sub(B, 8000)
d_a = torch.as_tensor(a, device="cuda"); d_b = torch.as_tensor(b, device="cuda"); torch.cuda.synchronize()
def call(iters):
    cfg = capi.Context.make_config(max_iterations=iters, tolerance=0.0, min_error=0.0)
    t0 = time.perf_counter(); r, h = ctx.align_device(d_b.data_ptr(), b.shape[0], d_a.data_ptr(), a.shape[0], cfg)
    return time.perf_counter() - t0, h
call(20)
t20 = min(call(20)[0] for _ in range(12)); t60 = min(call(60)[0] for _ in range(12))
out["points"] = [int(b.shape[0]), int(a.shape[0])]
out["period_us"] = round(1e6 * (t60 - t20) / 40, 3)
out["call20_ms"] = round(1e3 * t20, 4)
out["hist60_tail"] = float.hex(call(60)[1][-1])
for keep in (500, 2000, 8000):
    x, y = sub(A, keep), sub(B, keep)
    cfg = capi.Context.make_config()
    ctx.align(y, x, cfg)
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); res, hist = ctx.align(y, x, cfg); ts.append(time.perf_counter() - t0)
    out["frame_%%d" %% x.shape[0]] = {"align_ms": round(1e3 * min(ts), 4), "iterations": res.num_iterations, "final_error": float.hex(res.final_error)}
drive = os.environ["ICPMI_AB_DRIVE"]
paths = [p for _, p in capi.discover_frames(drive)]
odometry.run_odometry_stream(paths[:4], ctx)
best = None
for _ in range(3):
    t0 = time.perf_counter(); tr = odometry.run_odometry_stream(paths, ctx); w = time.perf_counter() - t0
    best = w if best is None or w < best else best
out["stream"] = {"frames": len(paths), "ms_per_frame": round(1e3 * best / (len(paths) - 1), 4), "median_ms": round(float(np.median(tr.frame_ms)), 4),
                 "iterations": int(sum(tr.iterations))}
print(json.dumps(out))
''' % (ROOT, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import run_sequence  # noqa: E402
drive = tempfile.mkdtemp(prefix="icpmi_ab_drive_")
run_sequence.write_synthetic_drive(drive, 0, 60, workers=min(16, os.cpu_count() or 1))
legs = []
for rep in range(3):
    for knob in ("0", "1"):
        env = dict(os.environ, ICPMI_SMALL=knob, ICPMI_AB_DRIVE=drive)
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        if r.returncode != 0:
            print(r.stderr[-3000:], file=sys.stderr)
            sys.exit(1)
        legs.append({"small": int(knob), **json.loads(r.stdout.strip().splitlines()[-1])})
def best(knob, f):
    return min(f(l) for l in legs if l["small"] == knob)
summary = {"period_us": {"general": best(0, lambda l: l["period_us"]), "small": best(1, lambda l: l["period_us"])},
           "stream_ms_per_frame": {"general": best(0, lambda l: l["stream"]["ms_per_frame"]), "small": best(1, lambda l: l["stream"]["ms_per_frame"])},
           "stream_median_ms": {"general": best(0, lambda l: l["stream"]["median_ms"]), "small": best(1, lambda l: l["stream"]["median_ms"])},
           "bits_equal": legs[0]["hist60_tail"] == legs[1]["hist60_tail"] and
                         all(legs[0][k]["final_error"] == legs[1][k]["final_error"] for k in legs[0] if k.startswith("frame_"))}
for k in legs[0]:
    if k.startswith("frame_"):
        summary[k + "_align_ms"] = {"general": best(0, lambda l: l[k]["align_ms"]), "small": best(1, lambda l: l[k]["align_ms"]),
                                    "iterations": [legs[0][k]["iterations"], legs[1][k]["iterations"]]}
print(json.dumps({"summary_min_over_legs": summary, "legs": legs}, indent=1))
