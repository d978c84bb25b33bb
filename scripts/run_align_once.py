"""One warm + a few timed icpmi_align_device calls on C3 (for rocprofv3 runs).
Usage: python scripts/run_align_once.py [engine] [n] [iterations] [calls]"""
import sys
import numpy as np
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidar_slam_from_scratch_amd import capi, synth

eng = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 30
calls = int(sys.argv[4]) if len(sys.argv) > 4 else 3
n_src = int(sys.argv[5]) if len(sys.argv) > 5 else n   # rows of the source used (one shard of it)
src, tgt, _ = synth.c3_uniform(n)
dsrc = torch.from_numpy(np.ascontiguousarray(src[:n_src])).cuda()
dtgt = torch.from_numpy(tgt).cuda()
cfg = capi.Context.make_config(iters, 0.0, 0.0)
ctx = capi.Context(device=0, search=eng, profile=0)
for _ in range(calls):
    res, hist = ctx.align_device(dsrc.data_ptr(), n_src, dtgt.data_ptr(), n, cfg)
torch.cuda.synchronize()
print(res.loop_iterations, res.final_error)
ctx.close()
