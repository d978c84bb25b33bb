#!/usr/bin/env python3
"""Secondary measurements (BASELINE.json configs[1], [3], [4] stand-ins): not the headline
bench.  Prints one JSON object."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import numpy as np
from lidar_slam_from_scratch_amd import capi, synth, odometry
from oracle import oracle as orc

out = {}


def timed_align(c, src, tgt, cfg, reps=5):
    """median wall time of `reps` calls (one sample right after a context's first call mostly measures the clocks coming up)"""
    ts, last = [], None
    for _ in range(reps):
        t0 = time.perf_counter(); last = c.align(src, tgt, cfg); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), last


ctx = capi.Context(device=0, profile=1)
pctx = capi.Context(device=0, search=capi.SEARCH_MFMA_BF16, profile=1)   # the all-pairs engine, same jobs (AUTO takes the culled one on targets of more than 16 splits, the small-cloud kernel below)

# C2 stand-in: LiDAR-like pair, ~20k points, reference defaults (50 it, tol 1e-6)
src, tgt, T = synth.c2_lidar_pair()
cfg = capi.Context.make_config()
ctx.align(src, tgt, cfg)
g, (res, hist) = timed_align(ctx, src, tgt, cfg)
t0 = time.perf_counter(); ref = orc.icp_point_to_plane(src, tgt); c = time.perf_counter() - t0
dt, dr = synth.pose_delta(np.array(res.transformation[:]).reshape(4, 4), ref.transformation)
pctx.align(src, tgt, cfg)
gp, (pres, _) = timed_align(pctx, src, tgt, cfg)
out["c2_lidar_pair"] = {"n_src": int(src.shape[0]), "n_tgt": int(tgt.shape[0]), "gpu_call_ms": 1e3 * g,
                        "gpu_call_ms_all_pairs_engine": 1e3 * gp, "all_pairs_iterations": pres.num_iterations,
                        "cpu_call_ms": 1e3 * c, "iterations": res.num_iterations, "pose_dt": dt, "pose_dr": dr,
                        "iters_equal": res.num_iterations == ref.num_iterations}

# the same pair at the size configs[1] names (~20k points: the stand-in at a 0.3 m voxel)
src, tgt, T = synth.c2_lidar_pair(voxel=0.3)
ctx.align(src, tgt, cfg)
g, (res, hist) = timed_align(ctx, src, tgt, cfg)
t0 = time.perf_counter(); ref = orc.icp_point_to_plane(src, tgt); c = time.perf_counter() - t0
dt, dr = synth.pose_delta(np.array(res.transformation[:]).reshape(4, 4), ref.transformation)
pctx.align(src, tgt, cfg)
gp, (pres, _) = timed_align(pctx, src, tgt, cfg)
out["c2_lidar_pair_20k"] = {"n_src": int(src.shape[0]), "n_tgt": int(tgt.shape[0]), "gpu_call_ms": 1e3 * g,
                            "gpu_call_ms_all_pairs_engine": 1e3 * gp, "cpu_call_ms": 1e3 * c, "iterations": res.num_iterations,
                            "pose_dt": dt, "pose_dr": dr, "iters_equal": res.num_iterations == ref.num_iterations}

# C5 stand-in: 12-frame synthetic drive, frame-to-frame odometry
frames = [synth.lidar_frame(f) for f in range(12)]
truth = [synth.lidar_pose(f) for f in range(12)]
odometry.run_odometry(frames[:3], odometry.gpu_align(ctx))
t0 = time.perf_counter(); tr = odometry.run_odometry(frames, odometry.gpu_align(ctx)); g = time.perf_counter() - t0
odometry.run_odometry(frames[:3], odometry.gpu_align(pctx))
t0 = time.perf_counter(); trp = odometry.run_odometry(frames, odometry.gpu_align(pctx)); gp = time.perf_counter() - t0
t0 = time.perf_counter()
rf = odometry.run_odometry(frames, lambda s, t, mi, tol: orc.icp_point_to_plane(s, t, mi, tol, 1e-9))
c = time.perf_counter() - t0
out["c5_odometry_12_frames"] = {"points_per_frame": int(np.mean([f.shape[0] for f in frames])),
                                "gpu_ms_per_frame": 1e3 * g / 11, "cpu_ms_per_frame": 1e3 * c / 11,
                                "gpu_ms_per_frame_all_pairs_engine": 1e3 * gp / 11,
                                "all_pairs_iterations_equal": trp.iterations == rf.iterations,
                                "ate_gpu_m": odometry.absolute_trajectory_error(tr, truth),
                                "ate_cpu_m": odometry.absolute_trajectory_error(rf, truth),
                                "iterations_equal": tr.iterations == rf.iterations}

# the same drive from RAW scans, clouds resident in HBM: upload, voxel filter and registration per frame
raw = [synth.lidar_frame(f, voxel=0) for f in range(12)]
odometry.run_odometry_device(raw[:3], ctx)
t0 = time.perf_counter(); trd = odometry.run_odometry_device(raw, ctx); g = time.perf_counter() - t0
t0 = time.perf_counter(); filt = [orc.voxel_downsample(r, 0.5) for r in raw]; cv = time.perf_counter() - t0
out["c5_device_resident_from_raw_scans"] = {
    "raw_points_per_frame": int(np.mean([r.shape[0] for r in raw])),
    "gpu_ms_per_frame_upload_voxel_icp": 1e3 * g / 11,
    "cpu_voxel_filter_ms_per_frame": 1e3 * cv / 12,
    "iterations": trd.iterations,
    "ate_gpu_m": odometry.absolute_trajectory_error(trd, truth)}

# C4: 1M -> 1M on one GPU, 3 iterations, against the oracle with all host cores
if "--c4" in sys.argv:
    src, tgt, T = synth.c4_uniform()
    cfg = capi.Context.make_config(3, 0.0, 0.0)
    t0 = time.perf_counter(); ctx.align(src, tgt, cfg); g_cold = time.perf_counter() - t0   # (allocates its workspaces)
    ctx.reset_profile()
    t0 = time.perf_counter(); res, hist = ctx.align(src, tgt, cfg); g = time.perf_counter() - t0
    p = ctx.get_profile()
    pctx.align(src[:200000], tgt, cfg)  # warm the workspaces
    t0 = time.perf_counter(); pres, phist = pctx.align(src, tgt, cfg); gp = time.perf_counter() - t0
    nth = os.cpu_count() or 8
    t0 = time.perf_counter(); ref = orc.icp_point_to_plane(src, tgt, 3, 0.0, 0.0, faithful=False, nthreads=nth); c = time.perf_counter() - t0
    dt, dr = synth.pose_delta(np.array(res.transformation[:]).reshape(4, 4), ref.transformation)
    out["c4_1M_3_iterations"] = {"gpu_call_s": g, "gpu_first_call_s_with_allocations": g_cold, "gpu_call_s_all_pairs_engine": gp,
                                 "all_pairs_hist_max_abs_diff": float(np.abs(phist - ref.error_history).max()),
                                 "cpu_call_s_%d_threads" % nth: c, "pose_dt": dt, "pose_dr": dr,
                                 "hist_max_abs_diff": float(np.abs(hist - ref.error_history).max()),
                                 "coarse_ms_per_pass": p["coarse_ms"] / max(p["coarse_launches"], 1),
                                 "pairs_culled_frac": 1.0 - p["nn_group_pairs_run"] / max(1, p["nn_group_pairs"])}
print(json.dumps(out, indent=1))
