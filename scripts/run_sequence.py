#!/usr/bin/env python3
"""Frame-to-frame odometry over a directory of scans (BASELINE.json configs[1] and [4]: KITTI
seq 00 when a copy is at hand, otherwise the synthetic drive of SURVEY section 8d written out in
KITTI's own .bin format).

    python scripts/run_sequence.py --data_dir /data/kitti/sequences/00/velodyne [--frames 0:200]
    python scripts/run_sequence.py --make-synthetic /tmp/drive --frames 0:200     # writes the .bin files first

What runs: discover_frames (file_utils.cpp:217-247) -> per frame icpmi_stream_push_file: file ->
pinned memory -> HBM (float32, widened on the device), voxel filter + min-points guard +
point-to-plane ICP against the previous filtered scan, which stays resident
(slam_node.cpp:121-152) -> the reference's gate
and pose update (slam_node.cpp:139-142).  With --oracle the same frames also go through the CPU
oracle loop (test infrastructure) for iteration counts and ATE.  Prints one JSON object.
A --data_dir that does not exist is skipped with a message and exit code 0 (no dataset ships with
the reference or this repo)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def _write_frame(job):
    out_dir, f, beams, azimuths = job
    from lidar_slam_from_scratch_amd import synth
    pts = synth.lidar_frame(f, voxel=0, beams=beams, azimuths=azimuths, **synth.DRIVE_200)
    rec = np.zeros((pts.shape[0], 4), dtype=np.float32)
    rec[:, :3] = pts
    rec.tofile(os.path.join(out_dir, "%06d.bin" % f))
    return f


def write_synthetic_drive(out_dir, first, last, beams=64, azimuths=1800, workers=1):
    """SURVEY 8d C5 stand-in: the C2 scene driven through (synth.DRIVE_200: 0.6 m per frame); raw scans as
    KITTI .bin (x, y, z, intensity float32) named like KITTI's velodyne files.  Returns ground truth.
    workers > 1: the frames are ray-cast by that many child processes of this script (`--write-frames`;
    numpy only, started fresh and never forked: the caller may hold a GPU context)."""
    from lidar_slam_from_scratch_amd import synth
    os.makedirs(out_dir, exist_ok=True)
    jobs = [(out_dir, f, beams, azimuths) for f in range(first, last)]
    workers = max(1, min(workers, len(jobs)))
    if workers > 1:
        import subprocess
        kids = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--write-frames", out_dir,
                                  "--frames", "%d:%d" % (first + w, last), "--frame-step", str(workers),
                                  "--beams", str(beams), "--azimuths", str(azimuths)]) for w in range(workers)]
        if any(k.wait() != 0 for k in kids):
            raise RuntimeError("a frame writer failed")
    else:
        for j in jobs:
            _write_frame(j)
    truth = [synth.lidar_pose(f, **synth.DRIVE_200) for f in range(first, last)]
    np.save(os.path.join(out_dir, "truth_poses.npy"), np.array(truth))
    return truth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data_dir", help="directory of .bin (KITTI) or .ply frames")
    ap.add_argument("--make-synthetic", metavar="DIR", help="write the synthetic drive there first and use it as data_dir")
    ap.add_argument("--frames", default=None, help="a:b slice of the discovered frames (synthetic: frame numbers)")
    ap.add_argument("--voxel", type=float, default=0.5)            # slam_node.cpp:18
    ap.add_argument("--max-iterations", type=int, default=50)      # slam_node.cpp:20
    ap.add_argument("--tolerance", type=float, default=1e-6)       # slam_node.cpp:21
    ap.add_argument("--min-points", type=int, default=1000)        # slam_node.hpp:30
    ap.add_argument("--search", type=int, default=0)
    ap.add_argument("--beams", type=int, default=64)
    ap.add_argument("--azimuths", type=int, default=1800)
    ap.add_argument("--oracle", action="store_true", help="also run the CPU oracle loop on the same frames (slow)")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="do not read the next frame's file on the library's worker thread while the current frame runs")
    ap.add_argument("--map", action="store_true",
                    help="per frame also the map side of process_frame (slam_node.cpp:147-153): world points of the resident scan "
                         "and the occupancy-grid insert, on the device; with --oracle the cell set is compared with the oracle's")
    ap.add_argument("--world-out", action="store_true", help="with --map: copy every frame's world points to the host (what :155 publishes)")
    ap.add_argument("--write-frames", metavar="DIR", help="(internal) only write the synthetic frames a:b:step there and exit")
    ap.add_argument("--frame-step", type=int, default=1)
    args = ap.parse_args()
    if args.write_frames:
        a, b = args.frames.split(":")
        for f in range(int(a), int(b), args.frame_step):
            _write_frame((args.write_frames, f, args.beams, args.azimuths))
        return 0
    lo, hi = 0, None
    if args.frames:
        a, b = args.frames.split(":")
        lo, hi = int(a or 0), (int(b) if b else None)
    truth = None
    if args.make_synthetic:
        truth = write_synthetic_drive(args.make_synthetic, lo, hi if hi is not None else lo + 200, args.beams, args.azimuths)
        args.data_dir, lo, hi = args.make_synthetic, 0, None
    if not args.data_dir or not os.path.isdir(args.data_dir):
        print("run_sequence: data_dir %r is absent: skipped (no dataset ships with the reference; see --make-synthetic)"
              % (args.data_dir,))
        return 0

    import torch  # noqa: F401
    from lidar_slam_from_scratch_amd import capi, odometry
    frames = capi.discover_frames(args.data_dir)[lo:hi]
    if len(frames) < 2:
        print("run_sequence: %d frame(s) in %s: nothing to register" % (len(frames), args.data_dir))
        return 0
    paths = [p for _, p in frames]
    tp = os.path.join(args.data_dir, "truth_poses.npy")
    if truth is None and os.path.exists(tp):
        truth = list(np.load(tp))[lo:hi]
    ctx = capi.Context(device=0, search=args.search)
    grid = capi.Context.make_grid_config() if args.map else None                       # slam_node.hpp:35-40
    odometry.run_odometry_stream(paths[:3], ctx, args.voxel, args.max_iterations, args.tolerance, args.min_points, grid,
                                 args.world_out, not args.no_prefetch)  # warm-up
    t0 = time.perf_counter()
    tr = odometry.run_odometry_stream(paths, ctx, args.voxel, args.max_iterations, args.tolerance, args.min_points, grid,
                                      args.world_out, not args.no_prefetch)
    wall = time.perf_counter() - t0
    out = {"data_dir": args.data_dir, "frames": len(paths), "first": frames[0][0], "last": frames[-1][0],
           "voxel": args.voxel, "max_iterations": args.max_iterations, "tolerance": args.tolerance,
           "next_file_prefetched": not args.no_prefetch,
           "gpu_ms_per_frame_file_to_pose": 1e3 * wall / (len(paths) - 1),
           "gpu_frame_ms_median": float(np.median(tr.frame_ms)), "gpu_frame_ms_p95": float(np.percentile(tr.frame_ms, 95)),
           "iterations_total": int(sum(tr.iterations)), "iterations_per_frame_mean": float(np.mean(tr.iterations)),
           "gated_frames": int(sum(tr.gated)), "not_converged": int(len(tr.converged) - sum(tr.converged)),
           "final_error_mean": float(np.nanmean(tr.final_errors)),
           "path_length_m": float(np.linalg.norm(np.diff(tr.positions(), axis=0), axis=1).sum()),
           "slowest_frames": [{"frame": int(i) + 1, "ms": round(float(tr.frame_ms[i]), 3), "iterations": int(tr.iterations[i])}
                              for i in np.argsort(tr.frame_ms)[::-1][:8]]}
    if truth is not None:
        out["ate_gpu_m"] = odometry.absolute_trajectory_error(tr, truth)
    gpu_cells = None
    if args.map:
        gpu_cells = ctx.occupancy_cells()
        out["map"] = {"occupied_cells": int(gpu_cells.shape[0]), "world_points_copied_out": bool(args.world_out),
                      "grid": {"resolution": grid.resolution, "height_min": grid.height_min, "height_max": grid.height_max,
                               "max_range": grid.max_range},
                      "note": "gpu_ms_per_frame_file_to_pose includes the map side of every frame"}
    if args.oracle:
        from oracle import oracle as orc
        nth = os.cpu_count() or 1
        t0 = time.perf_counter()
        clouds = (orc.voxel_downsample(capi.load_cloud(p), args.voxel) for p in paths)
        rf = odometry.run_odometry(clouds, lambda s, t, mi, tol: orc.icp_point_to_plane(s, t, mi, tol, 1e-9, nthreads=nth),
                                   args.max_iterations, args.tolerance, args.min_points)
        cw = time.perf_counter() - t0
        out["oracle"] = {"threads": nth, "cpu_ms_per_frame": 1e3 * cw / (len(paths) - 1),
                         "iterations_equal": tr.iterations == rf.iterations,
                         "iterations_differing_frames": int(sum(a != b for a, b in zip(tr.iterations, rf.iterations))),
                         "gates_equal": tr.gated == rf.gated,
                         "max_pose_dt_m": float(max(np.linalg.norm(a[:3, 3] - b[:3, 3]) for a, b in zip(tr.poses, rf.poses)))}
        if truth is not None:
            out["oracle"]["ate_cpu_m"] = odometry.absolute_trajectory_error(rf, truth)
        if args.map:
            # update_occupancy_grid (slam_node.cpp:211-221) by the oracle over the oracle-filtered scans, moved by the
            # GPU loop's poses (so that the comparison is of the map side alone); the first frame and the
            # too-few-points frames take no part, as in the reference (:69-76, :125-130)
            want = set()
            for i, p in enumerate(paths):
                curr = orc.voxel_downsample(capi.load_cloud(p), args.voxel)
                if i == 0 or curr.shape[0] < args.min_points:
                    continue
                T = tr.poses[i]
                R, t = T[:3, :3], T[:3, 3]
                world = np.stack([((curr[:, 0] * R[a, 0] + curr[:, 1] * R[a, 1]) + curr[:, 2] * R[a, 2]) + t[a] for a in range(3)], axis=1)
                orc.occupancy_update(want, world, t)
            out["oracle"]["occupied_cells"] = len(want)
            out["oracle"]["cell_sets_equal"] = want == set(map(tuple, gpu_cells.tolist()))
    print(json.dumps(out, indent=1))
    ctx.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
