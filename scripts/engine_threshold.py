#!/usr/bin/env python3
"""Where AUTO should change engines (run on the GPU box): registrations of n -> n point pairs, n from 6k to 100k, through
  small   the small-cloud kernel (icp_small.h; ICPMI_SMALL_MAX_SPLITS=16 so that it takes every size it can),
  general the all-pairs engine's general kernels (ICPMI_SMALL=0: k_nn_coarse_bounded + resolve + fused finish),
  culled  the culled engine (ICPMI_SEARCH_MFMA_PRUNED),
each in a child process of its own (the knobs are read once), device-resident clouds, two workloads: a LiDAR-like frame
pair thinned to n points (the caller's settings: 50 iterations, 1e-6 -- it converges in 4-7) and the uniform C3 cloud at n
points (10 forced iterations).  Prints one JSON object: ms per call by size and engine."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, json, os
sys.path.insert(0, %r)
import numpy as np, torch
from lidar_slam_from_scratch_amd import capi, synth
eng = int(sys.argv[1]); sizes = [int(x) for x in sys.argv[2].split(",")]
out = {}
A = synth.lidar_frame(0, voxel=0); B = synth.lidar_frame(1, voxel=0)
rng = np.random.default_rng(3)
ctx = capi.Context(device=0, search=eng)
def timed(src, tgt, cfg, reps=12):
    ds = torch.as_tensor(src, device="cuda"); dt = torch.as_tensor(tgt, device="cuda"); torch.cuda.synchronize()
    for _ in range(3):
        r, h = ctx.align_device(ds.data_ptr(), src.shape[0], dt.data_ptr(), tgt.shape[0], cfg)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r, h = ctx.align_device(ds.data_ptr(), src.shape[0], dt.data_ptr(), tgt.shape[0], cfg); ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts)), r.num_iterations
for n in sizes:
    a = np.ascontiguousarray(A[np.sort(rng.choice(A.shape[0], min(n, A.shape[0]), replace=False))])
    b = np.ascontiguousarray(B[np.sort(rng.choice(B.shape[0], min(n, B.shape[0]), replace=False))])
    ms, it = timed(b, a, capi.Context.make_config())
    s, t, _ = synth.c3_uniform(n)
    ms2, it2 = timed(s, t, capi.Context.make_config(10, 0.0, 0.0))
    out[str(n)] = {"lidar_ms": round(ms, 4), "lidar_iterations": it, "uniform10_ms": round(ms2, 4)}
print(json.dumps(out))
''' % ROOT
sizes = sys.argv[1] if len(sys.argv) > 1 else "6000,10000,14000,20000,28000,32000,40000,50000,70000,100000"
legs = {"small": (2, {"ICPMI_SMALL": "1", "ICPMI_SMALL_MAX_SPLITS": "16"}), "general": (2, {"ICPMI_SMALL": "0"}), "culled": (3, {})}
res = {}
for name, (eng, env) in legs.items():
    r = subprocess.run([sys.executable, "-c", CHILD, str(eng), sizes], env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        print(r.stderr[-2000:], file=sys.stderr); sys.exit(1)
    res[name] = json.loads(r.stdout.strip().splitlines()[-1])
table = {}
for n in sizes.split(","):
    table[n] = {k: {name: res[name][n][k] for name in legs} for k in ("lidar_ms", "uniform10_ms")}
    table[n]["splits"] = (int(n) + 2047) // 2048
    table[n]["lidar_iterations"] = res["general"][n]["lidar_iterations"]
print(json.dumps(table, indent=1))
