"""Soak of the frame stream's two threads and two streams: the 200-frame synthetic drive N times over on one context
(prefetch worker: read, upload, filter; caller: registration, next target, map side on every other run), every run's
poses, iteration counts and cell sets compared bit for bit with the first run's.  usage: python scripts/soak_stream.py [runs] [dir]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch  # noqa: F401
from lidar_slam_from_scratch_amd import capi, odometry
import run_sequence
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 30
d = sys.argv[2] if len(sys.argv) > 2 else "/tmp/drive_soak"
if not os.path.isdir(d) or len(capi.discover_frames(d)) < 200:
    run_sequence.write_synthetic_drive(d, 0, 200)
paths = [p for _, p in capi.discover_frames(d)][:200]
ctx = capi.Context(device=0)
grid = capi.Context.make_grid_config()
ref = {}
bad = 0
t0 = time.time()
for r in range(runs):
    with_map = r % 2 == 1
    tr = odometry.run_odometry_stream(paths, ctx, grid=grid if with_map else None, prefetch=(r % 3 != 2))
    cells = ctx.occupancy_cells() if with_map else None
    key = "map" if with_map else "plain"
    sig = (tr.iterations, [p.tobytes() for p in tr.poses], None if cells is None else cells.tobytes())
    if key not in ref:
        ref[key] = sig
    elif sig != ref[key]:
        bad += 1
        print("run %d (%s) differs from the first run of its kind" % (r, key))
assert ref["plain"][0] == ref["map"][0] and ref["plain"][1] == ref["map"][1]
print("soak: %d runs of %d frames, %d differing, %.1f s" % (runs, len(paths), bad, time.time() - t0))
ctx.close()
sys.exit(1 if bad else 0)
