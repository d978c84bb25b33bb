"""A few icpmi_align calls on the C2 stand-in (LiDAR-like pair, ~9k x 11k points at the 0.5 m voxel; ~18k x 20k at 0.3 m,
the size BASELINE.json configs[1] names), for rocprofv3.
Usage: python scripts/run_c2_once.py [engine] [calls] [voxel]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidar_slam_from_scratch_amd import capi, synth

eng = int(sys.argv[1]) if len(sys.argv) > 1 else 0
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 5
voxel = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
src, tgt, _ = synth.c2_lidar_pair(voxel=voxel)
dsrc = torch.from_numpy(src).cuda()
dtgt = torch.from_numpy(tgt).cuda()
cfg = capi.Context.make_config()
ctx = capi.Context(device=0, search=eng, profile=0)
for _ in range(calls):
    res, hist = ctx.align_device(dsrc.data_ptr(), src.shape[0], dtgt.data_ptr(), tgt.shape[0], cfg)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(20):
    res, hist = ctx.align_device(dsrc.data_ptr(), src.shape[0], dtgt.data_ptr(), tgt.shape[0], cfg)
print("ms per call", 1e3 * (time.perf_counter() - t0) / 20, "iterations", res.num_iterations, src.shape, tgt.shape)
ctx.close()
