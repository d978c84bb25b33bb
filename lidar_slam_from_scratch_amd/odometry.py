"""Frame-to-frame odometry driver: `SlamNode::process_frame` minus ROS
(slam_viz/src/ros/slam_node.cpp:118-157), the primary caller of the ICP hot path.

    curr = voxel_downsample(load(frame))                      :121-122  (caller supplies clouds)
    if curr.rows() < min_points: repeat last pose; prev = curr :125-130
    result = icp_point_to_plane(source=curr, target=prev, cfg) :132-138
    delta = identity if (!converged || final_error > 1.0)      :139-140
    new_pose = poses.back() * delta                            :142
    prev = curr                                                :152

`align` is any callable (source, target, max_iterations, tolerance) -> object with
.transformation (4x4), .converged, .final_error, .num_iterations -- the HIP library in the
product, the oracle in the parity tests.
"""
import time

import numpy as np


class OdometryTrack:
    def __init__(self):
        self.poses = [np.eye(4)]          # slam_node.cpp:63-64: pose 0 = identity
        self.deltas = []
        self.final_errors = []
        self.iterations = []
        self.converged = []
        self.gated = []                   # True where the identity fallback was taken
        self.frame_ms = []

    def positions(self):
        return np.array([p[:3, 3] for p in self.poses])


def run_odometry(frames, align, max_iterations=50, tolerance=1e-6, min_points=1000):
    """frames: iterable of N x 3 fp64 clouds in the sensor frame (already downsampled)."""
    track = OdometryTrack()
    prev = None
    for k, curr in enumerate(frames):
        curr = np.ascontiguousarray(curr, dtype=np.float64)
        if k == 0:
            prev = curr                    # slam_node.cpp:69-72
            continue
        t0 = time.perf_counter()
        if curr.shape[0] < min_points:     # slam_node.cpp:125-130
            track.poses.append(track.poses[-1].copy())
            track.deltas.append(np.eye(4))
            track.final_errors.append(float("nan"))
            track.iterations.append(0)
            track.converged.append(False)
            track.gated.append(True)
            prev = curr
            track.frame_ms.append(1e3 * (time.perf_counter() - t0))
            continue
        r = align(curr, prev, max_iterations, tolerance)
        bad = (not r.converged) or r.final_error > 1.0       # slam_node.cpp:139-140
        delta = np.eye(4) if bad else np.asarray(r.transformation, dtype=np.float64)
        track.poses.append(track.poses[-1] @ delta)           # slam_node.cpp:142
        track.deltas.append(delta)
        track.final_errors.append(r.final_error)
        track.iterations.append(r.num_iterations)
        track.converged.append(bool(r.converged))
        track.gated.append(bool(bad))
        prev = curr                                           # slam_node.cpp:152
        track.frame_ms.append(1e3 * (time.perf_counter() - t0))
    return track


def gpu_align(ctx):
    """Adapter: capi.Context -> the `align` callable above."""
    from . import capi

    class _R:
        pass

    def align(source, target, max_iterations, tolerance):
        cfg = capi.Context.make_config(max_iterations=max_iterations, tolerance=tolerance)
        res, _hist = ctx.align(source, target, cfg)
        r = _R()
        r.transformation = np.array(res.transformation[:]).reshape(4, 4)
        r.converged = bool(res.converged)
        r.final_error = res.final_error
        r.num_iterations = res.num_iterations
        return r

    return align


def run_odometry_device(raw_frames, ctx, voxel=0.5, max_iterations=50, tolerance=1e-6, min_points=1000):
    """The same loop with the clouds resident in HBM: each RAW scan is
    uploaded once, voxel-filtered on the device (slam_node.cpp:122 -> icpmi_voxel_downsample_device)
    and registered against the previous filtered scan, which never left the device
    (slam_node.cpp:132-133,152: the target of frame t+1 is the source of frame t).
    torch is used for device memory only."""
    import torch
    from . import capi
    track = OdometryTrack()
    prev = None       # (tensor, rows)
    cfg = capi.Context.make_config(max_iterations=max_iterations, tolerance=tolerance)
    for k, raw in enumerate(raw_frames):
        t0 = time.perf_counter()
        d_raw = torch.from_numpy(np.ascontiguousarray(raw, dtype=np.float64)).cuda()
        d_cur = torch.empty_like(d_raw)
        n = ctx.voxel_downsample_device(d_raw.data_ptr(), d_raw.shape[0], voxel, d_cur.data_ptr(), d_raw.shape[0])
        if k == 0:
            prev = (d_cur, n)              # slam_node.cpp:69-72
            continue
        if n < min_points:                 # slam_node.cpp:125-130
            track.poses.append(track.poses[-1].copy())
            track.deltas.append(np.eye(4))
            track.final_errors.append(float("nan"))
            track.iterations.append(0)
            track.converged.append(False)
            track.gated.append(True)
            prev = (d_cur, n)
            track.frame_ms.append(1e3 * (time.perf_counter() - t0))
            continue
        res, _hist = ctx.align_device(d_cur.data_ptr(), n, prev[0].data_ptr(), prev[1], cfg)
        bad = (not res.converged) or res.final_error > 1.0   # slam_node.cpp:139-140
        delta = np.eye(4) if bad else np.array(res.transformation[:]).reshape(4, 4)
        track.poses.append(track.poses[-1] @ delta)           # slam_node.cpp:142
        track.deltas.append(delta)
        track.final_errors.append(res.final_error)
        track.iterations.append(res.num_iterations)
        track.converged.append(bool(res.converged))
        track.gated.append(bool(bad))
        prev = (d_cur, n)                                     # slam_node.cpp:152
        track.frame_ms.append(1e3 * (time.perf_counter() - t0))
    return track


def run_odometry_stream(paths, ctx, voxel=0.5, max_iterations=50, tolerance=1e-6, min_points=1000, grid=None,
                        want_world=False, prefetch=True):
    """The same loop over frame FILES with everything but the file read on the device: one
    `icpmi_stream_push_file` per frame does slam_node.cpp:121-152 -- the scan goes from disk through
    pinned memory to HBM (`.bin`: float32 records, widened there), voxel filter, min-points guard,
    registration against the previous filtered scan that stayed resident -- and this function
    applies the reference's gate and pose update (slam_node.cpp:139-142).  With `grid` (a
    capi.GridConfig) every frame also gets the map side (slam_node.cpp:147-153, `icpmi_stream_map_update`):
    world points of the resident scan (copied out only with `want_world`) and the occupancy insert;
    track.cells then holds the size of the cell set after each frame.  With `prefetch` the next
    frame's file is read by the library's worker thread while this frame runs
    (`icpmi_stream_prefetch_file`).  No torch in here."""
    from . import capi
    track = OdometryTrack()
    track.cells = []
    cfg = capi.Context.make_config(max_iterations=max_iterations, tolerance=tolerance)
    ctx.stream_reset()
    if grid is not None:
        ctx.occupancy_clear()

    def map_side(n_rows):
        if grid is not None:                                      # slam_node.cpp:147-153
            _w, n_cells = ctx.stream_map_update(track.poses[-1], grid, want_world=want_world, n_rows=n_rows)
            track.cells.append(n_cells)

    paths = list(paths)
    for k, path in enumerate(paths):
        t0 = time.perf_counter()
        if prefetch and k + 1 < len(paths):
            ctx.stream_prefetch_file(paths[k + 1])                # read beside this frame's work
        res, _hist, info = ctx.stream_push_file(path, voxel, min_points, cfg)
        if info.status == capi.STREAM_FIRST_FRAME:                # slam_node.cpp:69-72: kept, not inserted into the grid
            continue
        if info.status == capi.STREAM_TOO_FEW_POINTS:            # slam_node.cpp:125-130
            track.poses.append(track.poses[-1].copy())
            track.deltas.append(np.eye(4))
            track.final_errors.append(float("nan"))
            track.iterations.append(0)
            track.converged.append(False)
            track.gated.append(True)
            track.frame_ms.append(1e3 * (time.perf_counter() - t0))   # (the reference returns before its map update here)
            continue
        bad = (not res.converged) or res.final_error > 1.0        # slam_node.cpp:139-140
        delta = np.eye(4) if bad else np.array(res.transformation[:]).reshape(4, 4)
        track.poses.append(track.poses[-1] @ delta)                # slam_node.cpp:142
        track.deltas.append(delta)
        track.final_errors.append(res.final_error)
        track.iterations.append(res.num_iterations)
        track.converged.append(bool(res.converged))
        track.gated.append(bool(bad))
        map_side(info.n_filtered)
        track.frame_ms.append(1e3 * (time.perf_counter() - t0))
    return track


def absolute_trajectory_error(track, truth_poses):
    """RMS translation error against ground-truth poses expressed relative to frame 0."""
    t0_inv = np.linalg.inv(truth_poses[0])
    err = [np.linalg.norm((t0_inv @ T)[:3, 3] - P[:3, 3]) for T, P in zip(truth_poses, track.poses)]
    return float(np.sqrt(np.mean(np.square(err))))
