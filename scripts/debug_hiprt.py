import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1]
def maps():
    libs = sorted({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l or 'libhsa-runtime' in l})
    print("loaded:", libs, flush=True)
import numpy as np
from lidar_slam_from_scratch_amd import capi, synth
if mode.startswith("libfirst"):
    capi.load_library()
import torch
maps()
src, tgt, _ = synth.c1_room_corner(1000)
prof = "prof" in mode
ctx = capi.Context(device=0, profile=prof)
print("ctx ok", flush=True)
idx, d2 = ctx.nearest_batch(tgt, src); print("nn ok", flush=True)
res, hist = ctx.align(src, tgt, capi.Context.make_config()); print("align ok", res.num_iterations, flush=True)
x = torch.ones(1000, device="cuda"); print("torch sum", float((x*2).sum()), flush=True)
t = torch.from_numpy(src).cuda(); tt = torch.from_numpy(tgt).cuda(); torch.cuda.synchronize()
res, hist = ctx.align_device(t.data_ptr(), 1000, tt.data_ptr(), 1000, capi.Context.make_config()); print("align_device ok", res.num_iterations, flush=True)
print(ctx.get_profile() if prof else "")
