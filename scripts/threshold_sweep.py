"""Where the MFMA engine starts to pay on small clouds: a registration (reference defaults) and a
normal estimation of LiDAR-like clouds of 500..8000 points with the exact fp64 engine (threshold
out of reach) and with the MFMA engine (threshold 64).  Each threshold runs in a child process (the threshold is read once per process)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, json, os
sys.path.insert(0, %r)
import numpy as np, torch
from lidar_slam_from_scratch_amd import capi, synth
out = {}
ctx = capi.Context(device=0)
A = synth.lidar_frame(3, voxel=0.5, **synth.DRIVE_200)
B = synth.lidar_frame(4, voxel=0.5, **synth.DRIVE_200)
rng = np.random.default_rng(0)
for keep in (300, 500, 1000, 1500, 2000, 3000, 4000, 6000, 8000):
    a = np.ascontiguousarray(A[np.sort(rng.choice(A.shape[0], min(keep, A.shape[0]), replace=False))])
    b = np.ascontiguousarray(B[np.sort(rng.choice(B.shape[0], min(keep, B.shape[0]), replace=False))])
    cfg = capi.Context.make_config()
    ctx.align(b, a, cfg)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); res, _ = ctx.align(b, a, cfg); ts.append(time.perf_counter() - t0)
    t0 = time.perf_counter(); ctx.estimate_normals(a, 20); tn = time.perf_counter() - t0
    out["%%d->%%d" %% (b.shape[0], a.shape[0])] = {"align_ms": round(1e3 * min(ts), 3), "iterations": res.num_iterations,
                                               "normals_ms": round(1e3 * tn, 3)}
print(json.dumps(out))
''' % ROOT
res = {}
for thr in (100000000, 64):
    env = dict(os.environ, ICPMI_MFMA_MIN_TARGETS=str(thr))
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    res[str(thr)] = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {"error": r.returncode}
print(json.dumps(res, indent=1))
