"""Time icpmi_align_device on the C3 workload with each search engine (device-resident inputs).
Usage: python scripts/engine_compare.py [n] [iterations]"""
import sys, time, json
import numpy as np
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidar_slam_from_scratch_amd import capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
src, tgt, _ = synth.c3_uniform(n)
dsrc = torch.from_numpy(src).cuda()
dtgt = torch.from_numpy(tgt).cuda()
cfg = capi.Context.make_config(iters, 0.0, 0.0)
out = {}
ref = None
for name, eng in (("mfma_bf16", capi.SEARCH_MFMA_BF16), ("mfma_pruned", capi.SEARCH_MFMA_PRUNED)):
    ctx = capi.Context(device=0, search=eng, profile=1)
    for _ in range(2):
        res, hist = ctx.align_device(dsrc.data_ptr(), n, dtgt.data_ptr(), n, cfg)
    torch.cuda.synchronize()
    ctx.reset_profile()
    t0 = time.perf_counter()
    reps = 5 if n <= 200_000 else 2
    for _ in range(reps):
        res, hist = ctx.align_device(dsrc.data_ptr(), n, dtgt.data_ptr(), n, cfg)
    dt = (time.perf_counter() - t0) / reps
    p = ctx.get_profile()
    T = np.array(res.transformation[:]).reshape(4, 4)
    if ref is None:
        ref = (T, hist.copy())
    out[name] = {"ms_per_call": dt * 1e3, "it_per_s": res.loop_iterations / dt,
                 "loop_ms": p["loop_ms"] / reps, "coarse_ms_avg": p["coarse_ms"] / max(1, p["coarse_launches"]),
                 "pruned_frac": p["nn_pruned_blocks"] / max(1, p["nn_coarse_blocks"]),
                 "pairs_culled_frac": (1.0 - p["nn_group_pairs_run"] / p["nn_group_pairs"]) if p["nn_group_pairs"] else 0.0,
                 "coarse_minima_bytes": int(p["coarse_minima_bytes"]),
                 "pose_delta_vs_first": synth.pose_delta(T, ref[0]),
                 "hist_maxdiff": float(np.abs(hist - ref[1]).max())}
    ctx.close()
print(json.dumps(out, indent=1))
