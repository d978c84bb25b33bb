import os, sys, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch.multiprocessing as mp
from lidar_slam_from_scratch_amd import capi, synth

def child(rank):
    from lidar_slam_from_scratch_amd import capi, synth
    src, tgt, _ = synth.c1_room_corner(1000)
    ctx = capi.Context(device=0)
    res, hist = ctx.align(src, tgt, capi.Context.make_config())
    print("child", rank, res.num_iterations, flush=True)
    ctx.close()

if __name__ == "__main__":
    mode = sys.argv[1]
    src, tgt, _ = synth.c1_room_corner(1000)
    ctx = capi.Context(device=0, profile=True)
    if mode == "warm":
        res, hist = ctx.align(src, tgt, capi.Context.make_config()); print("parent pre", res.num_iterations, flush=True)
    mp.spawn(child, nprocs=2, join=True)
    print("children done", flush=True)
    res, hist = ctx.align(src, tgt, capi.Context.make_config()); print("parent post", res.num_iterations, flush=True)
