"""SURVEY section 8f row N3's map side on the GPU: world points and the occupancy-grid insert
(slam_viz/src/ros/slam_node.cpp:147-153, :211-221) through the C ABI against the oracle's restatement.
Integer work: the cell sets must be EQUAL; world points bit for bit."""
import os
import sys

import numpy as np
import pytest
import torch

from lidar_slam_from_scratch_amd import capi, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


@pytest.fixture()
def ctx():
    c = capi.Context(device=0)
    yield c
    c.close()


def as_set(cells):
    return set(map(tuple, np.asarray(cells).tolist()))


def test_occupancy_update_equals_oracle_set(ctx, oracle):
    rng = np.random.default_rng(3)
    want = set()
    for trial in range(12):
        n = int(rng.integers(1, 20000))
        w = rng.normal(0, 25, (n, 3))
        w[:, 2] = rng.uniform(-1, 3, n)
        sensor = rng.normal(0, 5, 3)
        res = float(rng.choice([0.05, 0.2, 0.5]))
        grid = dict(resolution=res, height_min=0.3, height_max=2.0, max_range=float(rng.choice([10.0, 40.0])))
        oracle.occupancy_update(want, w, sensor, **grid)
        n_cells = ctx.occupancy_update(w, sensor, capi.Context.make_grid_config(**grid))
        assert n_cells == len(want)
    got = ctx.occupancy_cells()
    assert got.shape == (len(want), 2) and as_set(got) == want
    # sorted by x then y, no duplicates
    key = got[:, 0].astype(np.int64) * (1 << 32) + (got[:, 1].astype(np.int64) + (1 << 31))
    assert (np.diff(key) > 0).all()
    # an update with nothing new leaves the set as it is; clearing empties it (slam_node.cpp:224)
    assert ctx.occupancy_update(w, sensor, capi.Context.make_grid_config(**grid)) == len(want)
    ctx.occupancy_clear()
    assert ctx.occupancy_cells().shape == (0, 2)
    assert ctx.occupancy_update(np.zeros((0, 3)), [0, 0, 0]) == 0


def test_occupancy_boundaries_and_specials(ctx, oracle):
    s = np.array([1.0, -2.0, 0.0])
    w = np.array([
        [5.0, 5.0, 0.3], [5.0, 5.2, 2.0], [5.0, 5.4, np.nextafter(0.3, 0)], [5.0, 5.6, np.nextafter(2.0, 3)],
        [1.5, -2.0, 1.0], [np.nextafter(1.5, 0), -2.0, 1.0], [41.0, -2.0, 1.0], [np.nextafter(41.0, 50), -2.0, 1.0],
        [-0.2, -0.2, 1.0], [-0.2000001, 3.0, 1.0], [0.6, 0.6000000000000001, 1.0],
        [np.nan, 1.0, 1.0], [3.0, np.inf, 1.0], [3.0, 3.0, np.nan], [1e300, 0.0, 1.0], [4e8, 4e8, 1.0],
        [-4e8, 1.0, 1.0], [5.0, 5.0, 1.0], [5.01, 5.01, 1.5],
    ])
    for res, rmax in ((0.2, 40.0), (0.2, 1e12), (1e-3, 1e12), (1e-9, 1e300)):
        grid = dict(resolution=res, height_min=0.3, height_max=2.0, max_range=rmax)
        ctx.occupancy_clear()
        n = ctx.occupancy_update(w, s, capi.Context.make_grid_config(**grid))
        want = oracle.occupancy_update(set(), w, s, **grid)
        assert n == len(want) and as_set(ctx.occupancy_cells()) == want
    with pytest.raises(capi.IcpError):
        ctx.occupancy_update(w, s, capi.Context.make_grid_config(resolution=0.0))


def test_occupancy_from_device_memory_and_large(ctx, oracle):
    rng = np.random.default_rng(5)
    w = rng.uniform(-60, 60, (400_000, 3))
    w[:, 2] = rng.uniform(0, 2.5, w.shape[0])
    d = torch.from_numpy(w).cuda()
    torch.cuda.synchronize()
    n = ctx.occupancy_update_device(d.data_ptr(), w.shape[0], [0.5, 0.25, 0.0])
    want = oracle.occupancy_update(set(), w, [0.5, 0.25, 0.0])
    assert n == len(want) and as_set(ctx.occupancy_cells()) == want


def test_stream_map_update_is_the_map_side_of_process_frame(tmp_path, oracle):
    """20 frames of the synthetic drive through icpmi_stream_push_file, the reference's gate and pose update on
    the host, icpmi_stream_map_update per frame.  World points = the oracle-filtered scan moved by the same pose,
    bit for bit (same operation order as (curr * R^T).rowwise() + t^T, slam_node.cpp:147); the cell set = the
    oracle's update_occupancy_grid over those world points, frame after frame."""
    import run_sequence
    run_sequence.write_synthetic_drive(str(tmp_path), 0, 20, beams=32, azimuths=900)
    paths = [p for _, p in capi.discover_frames(str(tmp_path))]
    ctx = capi.Context(device=0)
    cfg = capi.Context.make_config()
    pose = np.eye(4)
    want_cells = set()
    grid = capi.Context.make_grid_config()
    for k, p in enumerate(paths):
        res, _, info = ctx.stream_push_file(p, 0.5, 1000, cfg)
        if info.status == capi.STREAM_REGISTERED:
            T = np.array(res.transformation[:]).reshape(4, 4)
            delta = np.eye(4) if (not res.converged or res.final_error > 1.0) else T          # slam_node.cpp:139-140
            pose = pose @ delta                                                               # :142
        world, n_cells = ctx.stream_map_update(pose, grid)
        curr = oracle.voxel_downsample(capi.load_cloud(p), 0.5)
        assert world.shape == curr.shape == (info.n_filtered, 3)
        assert (ctx.stream_current_scan() == curr).all()          # the resident filtered scan is the oracle's, bit for bit
        R, t = pose[:3, :3], pose[:3, 3]
        ref_world = np.empty_like(curr)
        for a in range(3):                          # ((x r_a0 + y r_a1) + z r_a2) + t_a, the product's index order
            ref_world[:, a] = ((curr[:, 0] * R[a, 0] + curr[:, 1] * R[a, 1]) + curr[:, 2] * R[a, 2]) + t[a]
        assert (world == ref_world).all()
        oracle.occupancy_update(want_cells, ref_world, t)
        assert n_cells == len(want_cells)
    assert as_set(ctx.occupancy_cells()) == want_cells and len(want_cells) > 500
    # without the grid, and without the copy-out
    w2, n2 = ctx.stream_map_update(pose, update_grid=False)
    assert (w2 == world).all() and n2 == len(want_cells)
    w3, n3 = ctx.stream_map_update(pose, grid, want_world=False)
    assert w3 is None and n3 == len(want_cells)
    ctx.stream_reset()
    with pytest.raises(capi.IcpError):
        ctx.stream_map_update(pose, grid)
    ctx.close()


def test_cpp_stream_mirror(tmp_path):
    """include/icp_mi355x.hpp's OdometryStream + OccupancyGrid from a plain C++17 program
    (tests/cpp/stream_demo.cpp) against the same loop driven from Python through the same C ABI:
    statuses, iteration counts, poses, world-point checksums, and the cell set must agree exactly."""
    import subprocess
    import run_sequence
    from lidar_slam_from_scratch_amd import build
    run_sequence.write_synthetic_drive(str(tmp_path), 0, 8, beams=32, azimuths=900)
    paths = [p for _, p in capi.discover_frames(str(tmp_path))]
    exe = tmp_path / "stream_demo"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "stream_demo.cpp"), "-o", str(exe), build.LIB_PATH,
                           "-Wl,-rpath," + os.path.dirname(build.LIB_PATH), "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib",
                           "-lamdhip64"])
    subprocess.check_call([str(exe), str(tmp_path / "o.f64")] + paths)
    o = np.fromfile(tmp_path / "o.f64")
    assert int(o[0]) == len(paths)
    ctx = capi.Context(device=0)
    cfg, grid = capi.Context.make_config(), capi.Context.make_grid_config()
    pose = np.eye(4)
    p = 1
    for path in paths:
        res, _, info = ctx.stream_push_file(path, 0.5, 1000, cfg)
        if info.status == capi.STREAM_REGISTERED:
            T = np.array(res.transformation[:]).reshape(4, 4)
            pose = pose @ (np.eye(4) if (not res.converged or res.final_error > 1.0) else T)
        world, n_cells = ctx.stream_map_update(pose, grid)
        status, iters, nf, nc = o[p:p + 4]
        assert (int(status), int(nf), int(nc)) == (info.status, info.n_filtered, n_cells)
        assert int(iters) == (res.num_iterations if info.status == capi.STREAM_REGISTERED else 0)
        cpose = o[p + 4:p + 20].reshape(4, 4)
        # the C++ 4x4 product adds in index order like numpy's matmul may not: compare to rounding
        np.testing.assert_allclose(cpose, pose, rtol=0, atol=1e-13)
        pose = cpose                                   # keep both loops on the same pose from here on
        assert abs(o[p + 20] - world.sum()) <= 1e-6 * max(1.0, abs(world).sum())
        p += 21
    cells = ctx.occupancy_cells()
    n = int(o[p]); p += 1
    assert n == cells.shape[0] and (o[p:p + 2 * n].reshape(n, 2) == cells).all()
    ctx.close()
