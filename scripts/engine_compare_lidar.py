"""Engines on a RAW (not voxel-filtered) LiDAR-like frame pair: ~100k points with the density
structure of a real scan (rings on the ground, walls, a 1/r^2 fall-off), reference defaults
(50 iterations, tolerance 1e-6).  Checks all-pairs vs pruned engine and times both.
Usage: python scripts/engine_compare_lidar.py [azimuths]"""
import os, sys, time, json
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidar_slam_from_scratch_amd import capi, synth

az = int(sys.argv[1]) if len(sys.argv) > 1 else 1800
src, tgt, _ = synth.c2_lidar_pair(voxel=0, beams=64, azimuths=az)
dsrc, dtgt = torch.from_numpy(np.ascontiguousarray(src)).cuda(), torch.from_numpy(np.ascontiguousarray(tgt)).cuda()
cfg = capi.Context.make_config()
out = {"n_src": int(src.shape[0]), "n_tgt": int(tgt.shape[0])}
ref = None
for name, eng in (("mfma_bf16", capi.SEARCH_MFMA_BF16), ("mfma_pruned", capi.SEARCH_MFMA_PRUNED)):
    ctx = capi.Context(device=0, search=eng, profile=1)
    for _ in range(2):
        res, hist = ctx.align_device(dsrc.data_ptr(), src.shape[0], dtgt.data_ptr(), tgt.shape[0], cfg)
    ctx.reset_profile()
    t0 = time.perf_counter()
    for _ in range(5):
        res, hist = ctx.align_device(dsrc.data_ptr(), src.shape[0], dtgt.data_ptr(), tgt.shape[0], cfg)
    dt = (time.perf_counter() - t0) / 5
    p = ctx.get_profile()
    T = np.array(res.transformation[:]).reshape(4, 4)
    if ref is None:
        ref = (T, hist.copy())
    out[name] = {"ms_per_call": 1e3 * dt, "iterations": res.num_iterations, "converged": bool(res.converged),
                 "units_culled_frac": p["nn_pruned_blocks"] / max(1, p["nn_coarse_blocks"]),
                 "pose_delta_vs_first": synth.pose_delta(T, ref[0]),
                 "hist_maxdiff": float(np.abs(hist - ref[1]).max()) if len(hist) == len(ref[1]) else None,
                 "nn_recheck_queries": p["nn_recheck_queries"], "nn_fallback_queries": p["nn_fallback_queries"],
                 "knn_fallback_rows": p["knn_fallback_rows"]}
    ctx.close()
print(json.dumps(out, indent=1))
