"""Cost of one occupancy update (7k world points) against the size of the cell set it goes into.
usage (GPU box): python scripts/occupancy_scaling.py [library.so]"""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from lidar_slam_from_scratch_amd import capi
if len(sys.argv) > 1:
    capi.load_library(sys.argv[1])
ctx = capi.Context(device=0)
rng = np.random.default_rng(0)
grid = capi.Context.make_grid_config(max_range=1e9)
out = {}
frame = rng.uniform(-30, 30, (7000, 3)); frame[:, 2] = 1.0
d_frame = torch.from_numpy(frame).cuda()
for target in (0, 20_000, 200_000, 2_000_000):
    ctx.occupancy_clear()
    n = 0
    while n < target:
        big = rng.uniform(-1, 1, (1_000_000, 3)) * (40.0 + 0.0005 * target); big[:, 2] = 1.0
        n = ctx.occupancy_update(big, [1e6, 1e6, 0.0], grid)
    torch.cuda.synchronize()
    ts = []
    for k in range(30):
        shifted = d_frame + torch.tensor([0.37 * k, 0.0, 0.0], device="cuda", dtype=torch.float64)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); n = ctx.occupancy_update_device(shifted.data_ptr(), 7000, [1e6, 1e6, 0.0], grid); ts.append(time.perf_counter() - t0)
    out[str(target)] = {"cells": n, "update_us_median": round(1e6 * float(np.median(ts)), 1)}
print(json.dumps(out))
