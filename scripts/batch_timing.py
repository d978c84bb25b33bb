"""Three loop-closure verifications (loop_closure.hpp:94-123: the newest scan against three older ones, 30
iterations, tolerance 1e-6) one after the other (icpmi_align x 3) and side by side (icpmi_align_batch) on one
GPU: min over repeats of each, and the ratio to ONE registration.  ~7k-point filtered frames of the synthetic
drive.  Prints one JSON object."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
from lidar_slam_from_scratch_amd import capi, synth
q = synth.lidar_frame(60, voxel=0.5, **synth.DRIVE_200)
tg = [synth.lidar_frame(f, voxel=0.5, **synth.DRIVE_200) for f in (59, 61, 58)]
ctx = capi.Context(device=0)
cfg = capi.Context.make_config(max_iterations=30, tolerance=1e-6)
def best(f, reps=15):
    f(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); out = f(); ts.append(time.perf_counter() - t0)
    return min(ts), out
t1, one = best(lambda: [ctx.align(q, tg[0], cfg)])
t3, seq = best(lambda: [ctx.align(q, t, cfg) for t in tg])
tb, bat = best(lambda: ctx.align_batch([q] * 3, tg, cfg))
same = all(tuple(a[0].transformation) == tuple(b[0].transformation) and (a[1] == b[1]).all() for a, b in zip(seq, bat))
print(json.dumps({"points": [int(q.shape[0])] + [int(t.shape[0]) for t in tg], "iterations": [r.num_iterations for r, _ in seq],
                  "one_ms": round(1e3 * t1, 4), "three_sequential_ms": round(1e3 * t3, 4), "three_batched_ms": round(1e3 * tb, 4),
                  "batched_over_one": round(tb / t1, 3), "batched_over_sequential": round(tb / t3, 3), "bit_identical": bool(same)}))
