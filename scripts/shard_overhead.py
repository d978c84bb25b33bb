"""What one rank of an N-way sharded C3 job does, timed on one GPU: n/N source rows against
the full 100k target, through (a) the plain path and (b) the sharded path with a 1-rank RCCL
communicator (k_finish + ncclAllReduce + k_step instead of the fused k_finish_step; normals
row slice + in-place all-gather).  No xGMI latency in here: a lower bound of the N-rank time.
Usage: python scripts/shard_overhead.py [engine]"""
import sys, time, json
import numpy as np
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidar_slam_from_scratch_amd import capi, synth

eng = int(sys.argv[1]) if len(sys.argv) > 1 else 0
src, tgt, _ = synth.c3_uniform(100_000)
dtgt = torch.from_numpy(tgt).cuda()
cfg = capi.Context.make_config(30, 0.0, 0.0)
out = {}
for world in (1, 2, 4, 8):
    n = 100_000 // world
    dsrc = torch.from_numpy(np.ascontiguousarray(src[:n])).cuda()
    for mode in ("plain", "rccl_1rank"):
        ctx = capi.Context(device=0, search=eng, profile=1)
        if mode == "rccl_1rank":
            ctx.comm_init(1, 0, ctx.comm_unique_id())
        for _ in range(2):
            res, hist = ctx.align_device(dsrc.data_ptr(), n, dtgt.data_ptr(), 100_000, cfg)
        torch.cuda.synchronize()
        ctx.reset_profile()
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            res, hist = ctx.align_device(dsrc.data_ptr(), n, dtgt.data_ptr(), 100_000, cfg)
        dt = (time.perf_counter() - t0) / reps
        p = ctx.get_profile()
        out["n=%d %s" % (n, mode)] = {"ms_per_call": round(dt * 1e3, 3), "it_per_s": round(res.loop_iterations / dt, 1),
                                      "loop_us_per_pass": round(1e3 * p["loop_ms"] / reps / 31, 1),
                                      "coarse_us": round(1e3 * p["coarse_ms"] / max(1, p["coarse_launches"]), 1)}
        if mode == "rccl_1rank":
            ctx.comm_finalize()
        ctx.close()
print(json.dumps(out, indent=1))
