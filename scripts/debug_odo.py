import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np
from lidar_slam_from_scratch_amd import capi, synth, odometry
frames = [synth.lidar_frame(f, beams=32, azimuths=900) for f in range(5)]
frames.insert(3, frames[2][:500])
print([f.shape for f in frames], flush=True)
for search in (1, 2):
    ctx = capi.Context(device=0, search=search, profile=2)
    for k in range(1, len(frames)):
        src, tgt = frames[k], frames[k - 1]
        if src.shape[0] < 1000:
            continue
        print("engine", search, "frame", k, src.shape, tgt.shape, flush=True)
        res, hist = ctx.align(src, tgt, capi.Context.make_config())
        print("  ->", res.converged, res.num_iterations, res.final_error, flush=True)
    ctx.close()
