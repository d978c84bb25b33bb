"""Normal estimation (icpmi_estimate_normals, host pointers: the median call time includes the upload, the Morton pre-pass and
the download, the same on both sides) on the clouds the callers bring: the 100k uniform cloud, a raw
LiDAR-like scan, a voxel-filtered one; round 2's slot-minimum form (ICPMI_KNN_LISTS=0) and the list form, alternating
child processes on one box.  Prints one JSON object."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, sys, time
import numpy as np
sys.path.insert(0, %r)
from lidar_slam_from_scratch_amd import capi, synth
clouds = {"c3_100k": synth.c3_uniform(100000)[1], "lidar_raw": synth.lidar_frame(0, voxel=0), "lidar_filtered": synth.lidar_frame(0)}
out = {}
for name, pts in clouds.items():
    ctx = capi.Context(device=0, profile=1)
    ctx.estimate_normals(pts, 20)
    ctx.reset_profile()
    reps, ts = 7, []
    for _ in range(reps):
        t0 = time.perf_counter()
        nrm = ctx.estimate_normals(pts, 20)
        ts.append((time.perf_counter() - t0) * 1e3)
    p = ctx.get_profile()
    out[name] = {"points": int(pts.shape[0]), "call_ms_median": sorted(ts)[reps // 2], "fallback_rows": int(p["knn_fallback_rows"]) // reps,
                 "checksum": float(np.abs(nrm).sum())}
    ctx.close()
print(json.dumps(out))
''' % ROOT


def run(lists):
    env = dict(os.environ, ICPMI_KNN_LISTS=str(lists))
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    if r.returncode:
        raise SystemExit(r.stderr[-2000:])
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


res = {"slot_minima": [], "lists": []}
for _ in range(2):
    res["slot_minima"].append(run(0))
    res["lists"].append(run(1))
a, b = res["slot_minima"][-1], res["lists"][-1]
res["same_normals"] = all(a[k]["checksum"] == b[k]["checksum"] for k in a)
print(json.dumps(res, indent=1))
