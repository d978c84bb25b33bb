"""What the HIP-event brackets of icpmi_options.profile cost on the C3 loop (wall per call)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidar_slam_from_scratch_amd import capi, synth
src, tgt, _ = synth.c3_uniform(100_000)
dsrc, dtgt = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()
torch.cuda.synchronize()
cfg = capi.Context.make_config(20, 0.0, 0.0)
for prof in (0, 1, 2, 0, 1, 2):
    ctx = capi.Context(device=0, profile=prof)
    ctx.align_device(dsrc.data_ptr(), 100_000, dtgt.data_ptr(), 100_000, cfg)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        ctx.align_device(dsrc.data_ptr(), 100_000, dtgt.data_ptr(), 100_000, cfg)
        ts.append(time.perf_counter() - t0)
    print("profile", prof, "ms per call: min %.3f median %.3f" % (1e3 * min(ts), 1e3 * float(np.median(ts))))
    ctx.close()
