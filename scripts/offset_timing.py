#!/usr/bin/env python3
"""What a badly initialised registration costs (rows whose list overflows take the resolve's exhaustive search):
C3 (100k -> 100k, nearest-neighbour spacing ~1 m) started 0 / 0.5 / 1.5 / 3 / 10 m off, 10 forced iterations, through the
default engine (culled), the all-pairs engine with bounded passes and the all-pairs engine with round 2's unbounded passes
(ICPMI_NN_BOUNDED=0), each in a child process; ms per call, rows searched exhaustively, and that the three agree.
Run on the GPU box; prints one JSON object."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, json
sys.path.insert(0, %r)
import numpy as np, torch
from lidar_slam_from_scratch_amd import capi, synth
eng = int(sys.argv[1])
src, tgt, _ = synth.c3_uniform(100000)
ds = torch.as_tensor(src, device="cuda"); dt = torch.as_tensor(tgt, device="cuda"); torch.cuda.synchronize()
ctx = capi.Context(device=0, search=eng, profile=2)
out = {}
for off in (0.0, 0.5, 1.5, 3.0, 10.0):
    T0 = np.eye(4); T0[:3, 3] = (off, 0.0, 0.0)
    cfg = capi.Context.make_config(max_iterations=10, tolerance=0.0, min_error=0.0, initial_transform=T0)
    ctx.align_device(ds.data_ptr(), src.shape[0], dt.data_ptr(), tgt.shape[0], cfg)
    ctx.reset_profile()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); r, h = ctx.align_device(ds.data_ptr(), src.shape[0], dt.data_ptr(), tgt.shape[0], cfg); ts.append(time.perf_counter() - t0)
    p = ctx.get_profile()
    out[str(off)] = {"ms": round(1e3 * float(np.median(ts)), 3), "exhaustive_rows_per_call": int(p["nn_fallback_queries"]) // 5,
                     "hist": [float.hex(v) for v in h]}
print(json.dumps(out))
''' % ROOT
legs = {"default (culled)": (0, {}), "all pairs, bounded": (2, {}), "all pairs, unbounded": (2, {"ICPMI_NN_BOUNDED": "0"})}
res = {}
for name, (eng, env) in legs.items():
    r = subprocess.run([sys.executable, "-c", CHILD, str(eng)], env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        print(r.stderr[-2000:], file=sys.stderr); sys.exit(1)
    res[name] = json.loads(r.stdout.strip().splitlines()[-1])
table = {}
for off in res["default (culled)"]:
    table[off + " m"] = {name: {"ms": res[name][off]["ms"], "exhaustive_rows": res[name][off]["exhaustive_rows_per_call"]} for name in legs}
    table[off + " m"]["histories_bit_equal"] = len({tuple(res[name][off]["hist"]) for name in legs}) == 1
print(json.dumps(table, indent=1))
