"""Parity of the HIP path (through the C ABI) against the CPU oracle and the committed
golden fixtures.  Bit-exact for indices, squared distances and normals; pose within
1e-4 m / 1e-4 rad (BASELINE.json north_star) -- in practice ~1e-15 -- with the same
iteration count and convergence flag.  Marked gpu: runs on the MI355X box only."""
import os
import zlib

import numpy as np
import pytest

from lidar_slam_from_scratch_amd import capi, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
POSE_TOL_M, POSE_TOL_RAD = 1e-4, 1e-4   # north_star tolerance
HIST_TOL = 1e-9


def crc(a):
    return np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()))


def T_of(res):
    return np.array(res.transformation[:]).reshape(4, 4)


def check_against(res, hist, ref_T, ref_conv, ref_iters, ref_hist):
    dt, dr = synth.pose_delta(T_of(res), ref_T)
    assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD, (dt, dr)
    assert bool(res.converged) == bool(ref_conv)
    assert res.num_iterations == int(ref_iters)
    assert len(hist) == len(ref_hist)
    np.testing.assert_allclose(hist, ref_hist, rtol=0, atol=HIST_TOL)
    assert res.final_error == hist[-1]
    return dt, dr


# ------------------------------------------------------------------ nearest neighbour
@pytest.mark.parametrize("n_qry,n_tgt,seed", [(1, 1, 0), (1, 2, 1), (7, 3, 2), (513, 255, 3),
                                               (512, 256, 4), (1000, 257, 5), (5000, 5000, 6),
                                               (20011, 19997, 7)])
def test_nearest_batch_bit_exact(gpu_ctx, oracle, n_qry, n_tgt, seed):
    rng = np.random.default_rng(seed)
    tgt = rng.uniform(-50, 50, (n_tgt, 3))
    qry = rng.uniform(-50, 50, (n_qry, 3))
    idx, d2 = gpu_ctx.nearest_batch(tgt, qry)
    oidx, od2 = oracle.KDTree(tgt).nearest_batch(qry)
    assert (idx == oidx).all()
    assert (d2 == od2).all()  # same fp64 operation order, no FMA


def test_nearest_batch_on_surfaces_and_self(gpu_ctx, oracle):
    src, tgt, _ = synth.c1_room_corner()
    idx, d2 = gpu_ctx.nearest_batch(tgt, src)
    oidx, od2 = oracle.KDTree(tgt).nearest_batch(src)
    assert (idx == oidx).all() and (d2 == od2).all()
    # idempotence: every target's nearest target is itself at distance 0
    idx, d2 = gpu_ctx.nearest_batch(tgt, tgt)
    assert (idx == np.arange(tgt.shape[0])).all() and (d2 == 0).all()


def test_nearest_batch_duplicates_pick_lowest_index(gpu_ctx):
    tgt = np.array([[1.0, 2, 3], [4, 5, 6], [1, 2, 3], [4, 5, 6]])
    qry = np.array([[1.0, 2, 3.1], [4, 5, 5.9]])
    idx, _ = gpu_ctx.nearest_batch(tgt, qry)
    assert list(idx) == [0, 1]


def test_nearest_batch_large_properties(gpu_ctx):
    """BASELINE.json full size (100k -> 100k): size-independent properties only."""
    src, tgt, _ = synth.c3_uniform()
    idx, d2 = gpu_ctx.nearest_batch(tgt, src)
    assert idx.min() >= 0 and idx.max() < tgt.shape[0]
    diff = tgt[idx] - src
    assert (((diff[:, 0] ** 2 + diff[:, 1] ** 2) + diff[:, 2] ** 2) == d2).all()
    rng = np.random.default_rng(0)
    for _ in range(8):  # no sampled target may be closer than the reported neighbour
        j = rng.integers(0, tgt.shape[0], size=src.shape[0])
        dj = ((tgt[j] - src) ** 2).sum(axis=1)
        assert (dj >= d2 * (1 - 1e-15)).all()
    from scipy.spatial import cKDTree
    _, ii = cKDTree(tgt).query(src[:20000])
    assert (idx[:20000] == ii).all()


# ------------------------------------------------------------------ normals
@pytest.mark.parametrize("k", [3, 8, 20, 33])
def test_normals_bit_exact(gpu_ctx, oracle, k):
    _, tgt, _ = synth.c1_room_corner(3000)
    got = gpu_ctx.estimate_normals(tgt, k)
    want = oracle.estimate_normals(tgt, None, k)
    assert np.abs(got - want).max() <= 1e-12
    assert (got == want).all(axis=1).mean() == 1.0


def test_normals_large_bit_exact(gpu_ctx, oracle):
    """Past the size where the MFMA coarse pass feeds the k-NN resolve."""
    _, tgt, _ = synth.c3_uniform(20000, seed=14, perm_seed=15)
    got = gpu_ctx.estimate_normals(tgt, 20)
    want = oracle.estimate_normals(tgt, None, 20, nthreads=4)
    assert (got == want).all()
    _, lid, _ = synth.c2_lidar_pair()
    assert (gpu_ctx.estimate_normals(lid, 20) == oracle.estimate_normals(lid, None, 20, nthreads=4)).all()


def test_normals_candidate_overflow_falls_back_exactly(gpu_ctx, oracle):
    """300 coincident points: more equidistant candidates than the resolve kernel's LDS
    list holds, so those rows must go through the exact fallback list -- same bits."""
    rng = np.random.default_rng(12)
    pts = np.concatenate([rng.uniform(-20, 20, (8700, 3)), np.tile([[1.5, -2.5, 0.25]], (300, 1))])
    got = gpu_ctx.estimate_normals(pts, 20)
    want = oracle.estimate_normals(pts, None, 20, nthreads=4)
    assert (got == want).all()
    assert (got[8700:] == [1.0, 0.0, 0.0]).all()  # zero covariance -> first axis


def test_normals_more_coincident_points_than_the_candidate_list(gpu_ctx, oracle):
    """700 coincident points: the in-kernel bound cannot be tightened below 0, the rows go to
    the exact workgroup-per-row kernel."""
    rng = np.random.default_rng(13)
    pts = np.concatenate([rng.uniform(-20, 20, (8700, 3)), np.tile([[1.5, -2.5, 0.25]], (700, 1))])
    got = gpu_ctx.estimate_normals(pts, 20)
    want = oracle.estimate_normals(pts, None, 20, nthreads=4)
    assert (got == want).all()


def test_normals_few_points(gpu_ctx):
    tgt = np.array([[0.0, 0, 0], [1, 0, 0]])
    assert (gpu_ctx.estimate_normals(tgt, 20) == [[0, 0, 1], [0, 0, 1]]).all()  # icp.hpp:34-37
    tri = np.array([[0.0, 0, 0], [1, 0, 0], [0, 1, 0]])
    n = gpu_ctx.estimate_normals(tri, 20)
    np.testing.assert_allclose(np.abs(n), [[0, 0, 1]] * 3, atol=1e-15)


def test_normals_golden_crc(gpu_ctx):
    g = np.load(os.path.join(GOLD, "c2_lidar_pair.npz"))
    _, tgt, _ = synth.c2_lidar_pair()
    assert crc(tgt) == g["tgt_crc"]
    got = gpu_ctx.estimate_normals(tgt, 20)
    assert (got[:32] == g["normals_head"]).all()
    assert crc(got) == g["normals_crc"]


# ------------------------------------------------------------------ solve / transform
def test_solve_point_to_plane(gpu_ctx, oracle):
    src, tgt, _ = synth.c1_room_corner()
    tree = oracle.KDTree(tgt)
    idx, _ = tree.nearest_batch(src)
    nrm = oracle.estimate_normals(tgt, tree, 20)
    got = gpu_ctx.solve_point_to_plane(src, tgt[idx], nrm[idx])
    want = oracle.solve_point_to_plane(src, tgt[idx], nrm[idx])
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)  # summation order differs
    assert (got[3] == [0, 0, 0, 1]).all()


def test_solve_single_plane_is_finite(gpu_ctx, oracle):
    rng = np.random.default_rng(9)
    src = np.c_[rng.uniform(-5, 5, (400, 2)), np.zeros(400)]
    nrm = np.tile([0.0, 0, 1.0], (400, 1))
    got = gpu_ctx.solve_point_to_plane(src, src + [0, 0, 0.01], nrm)
    assert np.isfinite(got).all() and abs(got[2, 3] - 0.01) < 1e-12
    assert got[0, 3] == 0 and got[1, 3] == 0


def test_transform_points_bit_exact(gpu_ctx):
    src, _, T = synth.c1_room_corner(4097)
    got = gpu_ctx.transform_points(T, src)
    x, y, z = src[:, 0], src[:, 1], src[:, 2]
    for r in range(3):
        want = ((x * T[r, 0] + y * T[r, 1]) + z * T[r, 2]) + T[r, 3]
        assert (got[:, r] == want).all()
    back = gpu_ctx.transform_points(synth.invert_transform(T), got)
    np.testing.assert_allclose(back, src, atol=1e-13)


# ------------------------------------------------------------------ full registration
CASES = {
    "c1_room_corner": lambda: synth.c1_room_corner(),
    "kat1_exact": lambda: synth.kat1_exact_pair(),
    "c2_lidar_pair": lambda: synth.c2_lidar_pair(),
    "c3_small_20k": lambda: synth.c3_uniform(20000, seed=14, perm_seed=15),
    "c3_uniform_100k": lambda: synth.c3_uniform(),
}


@pytest.mark.parametrize("name", list(CASES))
def test_align_matches_golden(gpu_ctx, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    src, tgt, _ = CASES[name]()
    assert crc(src) == g["src_crc"] and crc(tgt) == g["tgt_crc"], "generator drifted"
    cfg = capi.Context.make_config(int(g["max_iterations"]), float(g["tolerance"]), float(g["min_error"]))
    res, hist = gpu_ctx.align(src, tgt, cfg)
    dt, dr = check_against(res, hist, g["transformation"], g["converged"], g["num_iterations"],
                           g["error_history"])
    assert dt < 1e-9 and dr < 1e-9  # far inside the north_star tolerance


def test_align_matches_oracle_live(gpu_ctx, oracle):
    src, tgt, _ = synth.c1_room_corner(2500)
    for kw in (dict(max_iterations=50, tolerance=1e-6, min_error=1e-9),
               dict(max_iterations=3, tolerance=0.0, min_error=0.0),
               dict(max_iterations=1, tolerance=1e-6, min_error=1e-9),
               dict(max_iterations=0, tolerance=1e-6, min_error=1e-9)):
        ref = oracle.icp_point_to_plane(src, tgt, **kw)
        res, hist = gpu_ctx.align(src, tgt, capi.Context.make_config(**kw))
        check_against(res, hist, ref.transformation, ref.converged, ref.num_iterations, ref.error_history)


def test_align_initial_transform(gpu_ctx, oracle):
    src, tgt, T = synth.c1_room_corner(2500)
    T0 = synth.make_transform((0.005, -0.01, 0.02), (0.05, -0.02, 0.0))
    ref = oracle.icp_point_to_plane(src, tgt, initial_transform=T0)
    res, hist = gpu_ctx.align(src, tgt, capi.Context.make_config(initial_transform=T0))
    check_against(res, hist, ref.transformation, ref.converged, ref.num_iterations, ref.error_history)


def test_align_control_flow_invariants(gpu_ctx):
    """SURVEY section 8a R9 invariants."""
    _, tgt, _ = synth.c1_room_corner(1500)
    res, hist = gpu_ctx.align(tgt, tgt, capi.Context.make_config())  # min_error stop at iter 0
    assert res.converged and res.num_iterations == 1 and list(hist) == [0.0, 0.0]
    assert (T_of(res) == np.eye(4)).all()
    src, tgt, _ = synth.c1_room_corner()
    res, hist = gpu_ctx.align(src, tgt, capi.Context.make_config())  # tolerance stop
    k = res.loop_iterations - 1
    assert res.converged and len(hist) == k + 2 and res.num_iterations == k + 1
    assert res.final_error == hist[k]
    res, hist = gpu_ctx.align(src, tgt, capi.Context.make_config(4, 0.0, 0.0))  # exhaustion
    assert not res.converged and res.num_iterations == 4 and len(hist) == 5


def test_align_ragged_sizes_and_reuse(gpu_ctx, oracle):
    """Clouds of different sizes through one context (workspace reuse, odometry pattern
    slam_node.cpp:132-145: the target of call t+1 is the source of call t)."""
    frames = [synth.lidar_frame(f, beams=32, azimuths=600) for f in range(3)]
    for prev, cur in zip(frames[:-1], frames[1:]):
        ref = oracle.icp_point_to_plane(cur, prev)
        res, hist = gpu_ctx.align(cur, prev, capi.Context.make_config())
        check_against(res, hist, ref.transformation, ref.converged, ref.num_iterations, ref.error_history)


def test_align_run_to_run_bit_stable(gpu_ctx):
    src, tgt, _ = synth.c1_room_corner(3000)
    cfg = capi.Context.make_config()
    a, ha = gpu_ctx.align(src, tgt, cfg)
    b, hb = gpu_ctx.align(src, tgt, cfg)
    assert (T_of(a) == T_of(b)).all() and (ha == hb).all()


def test_host_mirror_api(gpu_ctx, oracle):
    import lidar_slam_from_scratch_amd as pkg
    src, tgt, _ = synth.c1_room_corner(2000)
    ref = oracle.icp_point_to_plane(src, tgt)
    r = pkg.icp_point_to_plane(pkg.PointCloud(src), pkg.PointCloud(tgt), pkg.ICPConfig(), ctx=gpu_ctx)
    assert r.converged == ref.converged and r.num_iterations == ref.num_iterations
    np.testing.assert_allclose(r.transformation.matrix(), ref.transformation, atol=1e-9)
    r2 = pkg.ICP(pkg.ICPConfig(max_iterations=30), ctx=gpu_ctx).align(src, tgt)
    assert r2.num_iterations == r.num_iterations and r2.success() == (r.converged and r.final_error < 0.1)
    nn = pkg.NearestNeighborSearch(tgt, ctx=gpu_ctx)
    matched, dist = nn.find_correspondences(src)
    oidx, od2 = oracle.KDTree(tgt).nearest_batch(src)
    assert (matched == tgt[oidx]).all() and (dist == np.sqrt(od2)).all()


# ------------------------------------------------------------------ voxel filter
@pytest.mark.parametrize("n,voxel,scale,offset", [(1, 0.5, 1.0, 0.0), (1000, 0.5, 10.0, 0.0),
                                                  (50000, 0.5, 30.0, 0.0), (50000, 0.25, 30.0, -700.0),
                                                  (20000, 2.0, 30.0, 1.0e5)])
def test_voxel_downsample_bit_exact(gpu_ctx, oracle, n, voxel, scale, offset):
    rng = np.random.default_rng(n)
    pts = rng.uniform(-scale, scale, (n, 3)) + offset
    got = gpu_ctx.voxel_downsample(pts, voxel)
    want = oracle.voxel_downsample(pts, voxel)
    assert got.shape == want.shape and (got == want).all()   # same voxels, same order, same bits


def test_voxel_downsample_lidar_frame_and_edge_cases(gpu_ctx, oracle):
    raw = synth.lidar_frame(0, voxel=None, beams=32, azimuths=900)
    got = gpu_ctx.voxel_downsample(raw, 0.5)
    assert (got == oracle.voxel_downsample(raw, 0.5)).all()
    # as a set it is what the reference's unordered_map version yields (numpy restatement)
    ref = synth.voxel_centroids(raw, 0.5)
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-12)
    # points exactly on voxel boundaries and negative coordinates: floor(), not truncation
    edge = np.array([[0.0, 0, 0], [0.5, 0, 0], [-0.5, 0, 0], [-1e-12, 0, 0], [0.4999999999, 0, 0], [1.0, 1.0, 1.0]])
    assert (gpu_ctx.voxel_downsample(edge, 0.5) == oracle.voxel_downsample(edge, 0.5)).all()
    assert (gpu_ctx.voxel_downsample(edge, 0.0) == edge).all()          # file_utils.cpp:152
    assert gpu_ctx.voxel_downsample(np.zeros((0, 3)), 0.5).shape == (0, 3)
    with pytest.raises(capi.IcpError) as e:                              # > 2^21 cells on an axis
        gpu_ctx.voxel_downsample(np.array([[0.0, 0, 0], [1e7, 0, 0]]), 1e-3)
    assert e.value.code == capi.ERR_ARG
    # a NaN or infinite coordinate is UB in the reference (cast of floor(NaN)): here it is an error, never a key that
    # lands in another voxel's bit fields; the context stays usable
    for bad in (np.nan, np.inf, -np.inf):
        for col in range(3):
            pts = np.array([[0.1, 0, 0], [0.3, 0.2, 0.1], [0.2, 0, 0]])
            pts[1, col] = bad
            with pytest.raises(capi.IcpError) as e:
                gpu_ctx.voxel_downsample(pts, 0.5)
            assert e.value.code == capi.ERR_ARG
    assert (gpu_ctx.voxel_downsample(edge, 0.5) == oracle.voxel_downsample(edge, 0.5)).all()
    # small and large grids: 13+13+13 bits of cell offsets, 14+14+13, 17+17+17, and one axis at its 21-bit limit
    rng = np.random.default_rng(77)
    for extent, voxel in (((4.0, 4.0, 4.0), 1e-3), ((8.0, 8.0, 4.0), 1e-3), ((100.0, 100.0, 100.0), 1e-3),
                          ((2000.0, 1.0, 1.0), 1e-3), ((50.0, 50.0, 3.0), 0.5)):
        pts = rng.uniform(0, 1, (6000, 3)) * np.array(extent)
        pts[0:5600:7] = pts[1:5600:7]                                    # some shared voxels
        got, want = gpu_ctx.voxel_downsample(pts, voxel), oracle.voxel_downsample(pts, voxel)
        assert got.shape == want.shape and (got == want).all()


def test_voxel_then_align_on_device(gpu_ctx, oracle):
    """The stream path of slam_node.cpp:121-138 without leaving HBM: raw scans -> voxel filter
    -> registration, all on device pointers."""
    import torch
    raw_t = synth.lidar_frame(0, voxel=None, beams=32, azimuths=900)
    raw_s = synth.lidar_frame(1, voxel=None, beams=32, azimuths=900)
    dev = torch.device("cuda", 0)
    d_t, d_s = torch.from_numpy(raw_t).to(dev), torch.from_numpy(raw_s).to(dev)
    o_t, o_s = torch.empty_like(d_t), torch.empty_like(d_s)
    torch.cuda.synchronize()
    n_t = gpu_ctx.voxel_downsample_device(d_t.data_ptr(), raw_t.shape[0], 0.5, o_t.data_ptr(), raw_t.shape[0])
    n_s = gpu_ctx.voxel_downsample_device(d_s.data_ptr(), raw_s.shape[0], 0.5, o_s.data_ptr(), raw_s.shape[0])
    res, hist = gpu_ctx.align_device(o_s.data_ptr(), n_s, o_t.data_ptr(), n_t, capi.Context.make_config())
    ref = oracle.icp_point_to_plane(oracle.voxel_downsample(raw_s, 0.5), oracle.voxel_downsample(raw_t, 0.5))
    check_against(res, hist, ref.transformation, ref.converged, ref.num_iterations, ref.error_history)


# ------------------------------------------------------------------ scan context + loop closure
def _off_sector_boundaries(cloud):
    """Drop points whose azimuth is within 1e-9 of a sector boundary: there the sector depends
    on the last ulp of atan2, which no two libms agree on (glibc on the CPU, OCML on the GPU;
    the synthetic scanner shoots rays exactly at multiples of 2*pi/900)."""
    q = (np.arctan2(cloud[:, 1], cloud[:, 0]) + np.pi) / (2 * np.pi / 60)
    return cloud[np.abs(q - np.round(q)) > 1e-9]


def test_scan_context_bit_exact(gpu_ctx, oracle):
    raw = [synth.lidar_frame(f, beams=32, azimuths=900) for f in (0, 1, 7, 30)]
    for f in raw:       # unfiltered: at most a boundary point or two may land in the next sector
        assert (gpu_ctx.scan_context(f) != oracle.scan_context(f)).sum() <= 2
    frames = [_off_sector_boundaries(f) for f in raw]
    descs = [gpu_ctx.scan_context(f) for f in frames]
    for f, d in zip(frames, descs):
        assert (d == oracle.scan_context(f)).all()
    rng = np.random.default_rng(0)
    wild = np.c_[rng.uniform(-120, 120, (5000, 2)), rng.uniform(-3, 20, 5000)]   # beyond 80 m, near 0.1 m
    wild[:50, :2] *= 1e-3
    wild = _off_sector_boundaries(wild)
    assert (gpu_ctx.scan_context(wild) == oracle.scan_context(wild)).all()
    assert (gpu_ctx.scan_context(np.zeros((0, 3))) == 0).all()
    hist = np.stack(descs[1:] + [np.zeros((20, 60))])
    got = gpu_ctx.scan_context_distances(descs[0], hist)
    want = [oracle.scan_context_distance(descs[0], h) for h in hist]
    assert (got == np.array(want)).all()
    assert got[-1] == 1.0                                     # zero-norm descriptor -> 1 (scan_context.hpp:138)
    assert gpu_ctx.scan_context_distances(descs[0], descs[0][None])[0] < 1e-15


def test_loop_closure_detector_matches_oracle(gpu_ctx, oracle):
    """LoopClosureDetector::detect (loop_closure.hpp:66-126) with the GPU behind it against
    the same host logic with the oracle behind it: a short drive that comes back to its
    start."""
    from lidar_slam_from_scratch_amd import loop_closure as lc

    class OracleBackend:
        def scan_context(self, cloud):
            return oracle.scan_context(cloud)

        def distances(self, q, hist):
            return np.array([oracle.scan_context_distance(q, h) for h in hist])

        def align(self, s, t, mi, tol):
            return oracle.icp_point_to_plane(s, t, mi, tol, 1e-9)

    order = [0, 2, 4, 6, 8, 10, 12, 1, 3]      # frames 1 and 3 revisit the start of the drive
    clouds = [_off_sector_boundaries(synth.lidar_frame(f, beams=32, azimuths=900)) for f in order]
    cfg = lc.LoopClosureConfig(frame_gap=5, sc_distance_threshold=0.2, icp_fitness_threshold=0.3)
    dets = [lc.LoopClosureDetector(lc.GpuBackend(gpu_ctx), cfg), lc.LoopClosureDetector(OracleBackend(), cfg)]
    found = [[], []]
    for k, c in enumerate(clouds):
        for d, out in zip(dets, found):
            d.add_frame(c, k)
            out.extend(d.detect())
    assert len(found[0]) == len(found[1]) and len(found[0]) >= 1
    for a, b in zip(found[0], found[1]):
        assert (a.query_frame, a.match_frame) == (b.query_frame, b.match_frame)
        assert a.scan_context_distance == b.scan_context_distance
        dt, dr = synth.pose_delta(a.transform, b.transform)
        assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD and abs(a.icp_fitness - b.icp_fitness) < 1e-9


def test_cpp_loop_closure_mirror(tmp_path):
    """include/icp_mi355x.hpp's ScanContext / LoopClosureDetector from a plain C++17 program
    (tests/cpp/loop_demo.cpp) against the Python mirror through the same C ABI: same closures, same
    Scan Context distances and ICP fitness, same transforms."""
    import subprocess
    from lidar_slam_from_scratch_amd import build, loop_closure as lc
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "loop_demo"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "loop_demo.cpp"), "-o", str(exe), build.LIB_PATH,
                           "-Wl,-rpath," + os.path.dirname(build.LIB_PATH), "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib",
                           "-lamdhip64"])
    order = [0, 2, 4, 6, 8, 10, 12, 1, 3]      # frames 1 and 3 revisit the start of the drive
    clouds = [_off_sector_boundaries(synth.lidar_frame(f, beams=32, azimuths=900)) for f in order]
    files = []
    for k, c in enumerate(clouds):
        c.tofile(tmp_path / ("c%d.f64" % k))
        files.append(str(tmp_path / ("c%d.f64" % k)))
    subprocess.check_call([str(exe), str(tmp_path / "o.f64"), "5", "0.2", "0.3"] + files)
    o = np.fromfile(tmp_path / "o.f64")
    gpu_ctx = capi.Context(device=0)            # the program's default context: search AUTO (exact comparisons below)
    det = lc.LoopClosureDetector(lc.GpuBackend(gpu_ctx), lc.LoopClosureConfig(frame_gap=5, sc_distance_threshold=0.2,
                                                                              icp_fitness_threshold=0.3))
    want = []
    for k, c in enumerate(clouds):
        det.add_frame(c, k)
        want.extend(det.detect())
    n = int(o[0])
    assert n == len(want) and n >= 1
    p = 1
    for w in want:
        assert (int(o[p]), int(o[p + 1])) == (w.query_frame, w.match_frame)
        assert o[p + 2] == w.scan_context_distance and o[p + 3] == w.icp_fitness
        assert (o[p + 4:p + 20].reshape(4, 4) == np.asarray(w.transform).reshape(4, 4)).all()
        p += 20
    assert o[p] == gpu_ctx.scan_context_distances(gpu_ctx.scan_context(clouds[-1]), gpu_ctx.scan_context(clouds[0])[None])[0]
    assert int(o[p + 1]) == len(clouds)
    gpu_ctx.close()


# ------------------------------------------------------------------ odometry stream
def test_odometry_stream_matches_oracle(gpu_ctx, oracle):
    """process_frame minus ROS (slam_node.cpp:118-157) over a short synthetic drive: same
    poses, gates and iteration counts as the oracle-driven loop; ATE vs ground truth."""
    from lidar_slam_from_scratch_amd import odometry

    frames = [synth.lidar_frame(f, beams=32, azimuths=900) for f in range(5)]
    frames.insert(3, frames[2][::11][:500])    # a too-small (but well spread) frame: min_points guard
    truth = [synth.lidar_pose(f) for f in (0, 1, 2, 2, 3, 4)]

    def oracle_align(src, tgt, max_it, tol):
        return oracle.icp_point_to_plane(src, tgt, max_it, tol, 1e-9)

    ref = odometry.run_odometry(frames, oracle_align)
    got = odometry.run_odometry(frames, odometry.gpu_align(gpu_ctx))
    assert got.iterations == ref.iterations and got.gated == ref.gated and got.converged == ref.converged
    assert got.gated[2] and got.iterations[2] == 0              # the small frame repeats the pose
    for a, b in zip(got.poses, ref.poses):
        dt, dr = synth.pose_delta(a, b)
        assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD
    assert abs(odometry.absolute_trajectory_error(got, truth)
               - odometry.absolute_trajectory_error(ref, truth)) < 1e-9


def test_device_resident_odometry_matches_oracle(gpu_ctx, oracle):
    """The stream with the clouds kept in HBM: raw scan -> voxel filter on the device ->
    registration against the previous filtered scan (device pointers throughout).  The oracle
    loop filters with orc_voxel_downsample (same key order, bit-identical centroids) and runs
    the CPU ICP: same iteration counts, gates and poses."""
    from lidar_slam_from_scratch_amd import odometry

    raw = [synth.lidar_frame(f, voxel=0, beams=32, azimuths=900) for f in range(4)]
    raw.insert(2, raw[1][::40][:700])          # a scan too small after filtering: min_points guard

    def oracle_align(src, tgt, max_it, tol):
        return oracle.icp_point_to_plane(src, tgt, max_it, tol, 1e-9)

    ref = odometry.run_odometry([oracle.voxel_downsample(r, 0.5) for r in raw], oracle_align)
    got = odometry.run_odometry_device(raw, gpu_ctx, voxel=0.5)
    assert got.iterations == ref.iterations and got.gated == ref.gated and got.converged == ref.converged
    assert got.gated[1] and got.iterations[1] == 0
    for a, b in zip(got.poses, ref.poses):
        dt, dr = synth.pose_delta(a, b)
        assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD


# ------------------------------------------------------------------ error behaviour
def test_error_codes(gpu_ctx):
    cfg = capi.Context.make_config()
    pts = np.zeros((4, 3))
    empty = np.zeros((0, 3))
    with pytest.raises(capi.IcpError) as e:
        gpu_ctx.align(empty, pts, cfg)
    assert e.value.code == capi.ERR_EMPTY_SOURCE
    with pytest.raises(capi.IcpError) as e:
        gpu_ctx.align(pts, empty, cfg)
    assert e.value.code == capi.ERR_EMPTY_TARGET
    with pytest.raises(capi.IcpError) as e:
        gpu_ctx.nearest_batch(empty, pts)
    assert e.value.code == capi.ERR_EMPTY_TARGET
    with pytest.raises(capi.IcpError) as e:
        gpu_ctx.estimate_normals(pts, 65)
    assert e.value.code == capi.ERR_ARG
    import ctypes as C
    res = capi.Result()
    hist = np.zeros(4)
    rc = gpu_ctx._lib.icpmi_align(gpu_ctx._h, pts.ctypes.data_as(C.POINTER(C.c_double)), 4,
                                  pts.ctypes.data_as(C.POINTER(C.c_double)), 4, C.byref(cfg),
                                  C.byref(res), hist.ctypes.data_as(C.POINTER(C.c_double)), 4)
    assert rc == capi.ERR_CAPACITY and b"needs 51" in gpu_ctx._lib.icpmi_last_error(gpu_ctx._h)
    rc = gpu_ctx._lib.icpmi_align(gpu_ctx._h, None, 4, pts.ctypes.data_as(C.POINTER(C.c_double)), 4,
                                  C.byref(cfg), C.byref(res), hist.ctypes.data_as(C.POINTER(C.c_double)), 51)
    assert rc == capi.ERR_NULL
    import lidar_slam_from_scratch_amd as pkg
    r = pkg.icp_point_to_plane(empty, pts, ctx=gpu_ctx, on_error="unconverged")
    assert not r.converged and (r.transformation.matrix() == np.eye(4)).all()  # gate -> identity
    # the context is still usable after errors
    idx, _ = gpu_ctx.nearest_batch(np.eye(3), np.eye(3))
    assert list(idx) == [0, 1, 2]


def test_mfma_engine_certificate_paths(gpu_ctx, oracle):
    """Inputs that force the resolve kernel's rare branches (a passing check on benign data
    never exercises them): many targets whose distances to the query differ by less than
    the fp32 bound -- in the same column slot, in other columns of the split, and in other
    splits -- plus queries far outside the target's bounding box."""
    rng = np.random.default_rng(77)
    m = 3 * 2048 + 77
    tgt = rng.uniform(-60, 60, (m, 3))
    qry = rng.uniform(-60, 60, (512, 3))
    # a shell of near-equidistant targets around the first 64 queries, spread over slots
    for q in range(64):
        dirs = rng.normal(size=(24, 3))
        dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        radius = 0.05 * (1.0 + 1e-7 * rng.uniform(-1, 1, size=(24, 1)))
        where = rng.choice(m, size=24, replace=False)
        tgt[where] = qry[q] + dirs * radius
    qry[64:96] *= 40.0   # far outside the box: large |P| -> large bound
    idx, d2 = gpu_ctx.nearest_batch(tgt, qry)
    oidx, od2 = oracle.nearest_batch_brute(tgt, qry)
    assert (idx == oidx).all() and (d2 == od2).all()
    if gpu_ctx.engine.startswith("mfma"):
        gpu_ctx.reset_profile()
        gpu_ctx.nearest_batch(tgt, qry)
        p = gpu_ctx.get_profile()
        assert p["nn_recheck_queries"] + p["nn_fallback_queries"] > 0  # the branches ran


@pytest.mark.parametrize("scale,offset,seed", [(1.0, 0.0, 1), (100.0, 0.0, 2), (0.01, 0.0, 3),
                                               (1.0, 1.0e4, 4), (30.0, -2.5e3, 5)])
def test_mfma_engine_near_ties_across_scales(gpu_ctx, oracle, scale, offset, seed):
    """The certificate's error bound must hold whatever the cloud's size and position:
    clustered targets, queries whose two best candidates differ by ~1e-9 relative, all at
    different scales and far from the origin."""
    rng = np.random.default_rng(seed)
    m, nq = 9000, 1500
    centers = rng.uniform(-40, 40, (60, 3))
    tgt = centers[rng.integers(0, 60, m)] + rng.normal(0, 1.5, (m, 3))
    qry = tgt[rng.integers(0, m, nq)] + rng.normal(0, 0.3, (nq, 3))
    # plant pairs of targets at (almost) the same distance from the first 400 queries
    for q in range(400):
        d = rng.normal(size=(2, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        r = 0.02 * (1.0 + np.array([0.0, 1e-9 * rng.uniform(-1, 1)]))
        tgt[rng.choice(m, 2, replace=False)] = qry[q] + d * r[:, None]
    tgt = tgt * scale + offset
    qry = qry * scale + offset
    idx, d2 = gpu_ctx.nearest_batch(tgt, qry)
    oidx, od2 = oracle.nearest_batch_brute(tgt, qry)
    assert (idx == oidx).all() and (d2 == od2).all()
    nrm = gpu_ctx.estimate_normals(tgt, 20)
    assert (nrm == oracle.estimate_normals(tgt, None, 20, nthreads=4)).all()


def test_mfma_engine_exact_ties(gpu_ctx, oracle):
    """Exact fp64 ties across slots and splits resolve to the lowest index."""
    rng = np.random.default_rng(5)
    base = rng.uniform(-10, 10, (5000, 3))
    tgt = np.concatenate([base, base[::-1], base])      # every point three times
    qry = base[:700] + 1e-3
    idx, d2 = gpu_ctx.nearest_batch(tgt, qry)
    oidx, od2 = oracle.nearest_batch_brute(tgt, qry)
    assert (idx == oidx).all() and (d2 == od2).all()
    assert (idx < 5000).all()


def test_nonfinite_input_is_memory_safe(gpu_ctx):
    """NaN / Inf coordinates are undefined behaviour in the reference (kdtree.hpp:33-36 ->
    row(-1)); here they must neither fault the GPU nor poison the context."""
    src, tgt, _ = synth.c3_uniform(9000, seed=31, perm_seed=32)
    bad = src.copy()
    bad[::7] = np.nan
    bad[3::11, 1] = np.inf
    idx, _ = gpu_ctx.nearest_batch(tgt, bad)
    assert (idx[::7] == -1).all()                      # nothing compares less (kdtree.hpp:53)
    ok = np.isfinite(bad).all(axis=1)
    from scipy.spatial import cKDTree
    assert (idx[ok] == cKDTree(tgt).query(bad[ok])[1]).all()
    res, hist = gpu_ctx.align(bad, tgt, capi.Context.make_config(5, 0.0, 0.0))
    assert not np.isfinite(res.final_error)
    tbad = tgt.copy()
    tbad[5] = np.nan
    gpu_ctx.align(src, tbad, capi.Context.make_config(3, 0.0, 0.0))   # NaN in the target
    gpu_ctx.estimate_normals(tbad, 20)
    # a tiny degenerate target after a large one (stale neighbour lists must not be gathered)
    gpu_ctx.align(src, tgt[:40] * 1e-3, capi.Context.make_config(3, 0.0, 0.0))
    res, _ = gpu_ctx.align(src, tgt, capi.Context.make_config(3, 0.0, 0.0))
    assert np.isfinite(res.final_error)                 # the context still works


# (110,000 and 300,000 targets: more slot minima per row than a lane keeps in registers -- 1,664 slots = 106,496
#  targets --, so the k-NN resolve's tail loops run: half a trip at 110k, 24 trips at 300k)
@pytest.mark.parametrize("m,k", [(700, 5), (700, 20), (12000, 1), (12000, 20), (12000, 32), (12000, 40),
                                 (110000, 20), (300000, 20)])
def test_k_nearest_matches_oracle(gpu_ctx, oracle, m, k):
    """icpmi_k_nearest = KDTree::k_nearest (kdtree.hpp:65-78) for a batch: closest first, the query
    point itself included when it is a target (distance 0).  Against the oracle's kd-tree k-NN
    and its brute-force form, for queries that are targets and queries that are not."""
    _, tgt, _ = synth.c3_uniform(m, seed=71 + m, perm_seed=72)
    rng = np.random.default_rng(5)
    qry = np.vstack([tgt[rng.choice(m, 40, replace=False)],                       # targets themselves
                     tgt[rng.choice(m, 40, replace=False)] + rng.normal(0, 0.3, (40, 3)),
                     rng.uniform(-60, 60, (20, 3))])                              # partly outside the cloud
    idx, d2 = gpu_ctx.k_nearest(tgt, qry, k)
    assert idx.shape == (100, k) and d2.shape == (100, k)
    tree = oracle.KDTree(tgt)
    for i in range(qry.shape[0]):
        want = oracle.k_nearest_brute(tgt, qry[i], k)
        assert list(idx[i]) == list(want), i
        assert list(tree.k_nearest(qry[i], k)) == list(want), i
        assert (d2[i] == ((tgt[want] - qry[i]) ** 2).sum(axis=1)).all() or np.allclose(d2[i], ((tgt[want] - qry[i]) ** 2).sum(axis=1), rtol=1e-15, atol=0)
        assert (np.diff(d2[i]) >= 0).all()
    assert (idx[:40, 0] == np.array([np.flatnonzero((tgt == q).all(axis=1))[0] for q in qry[:40]])).all()


def test_k_nearest_edge_cases(gpu_ctx):
    from lidar_slam_from_scratch_amd import icp
    tgt = np.array([[0.0, 0, 0], [1, 0, 0], [0, 2, 0], [0, 0, 3], [5, 5, 5]])
    idx, d2 = gpu_ctx.k_nearest(tgt, np.array([[0.1, 0, 0], [np.nan, 0, 0]]), 8)
    assert list(idx[0]) == [0, 1, 2, 3, 4, -1, -1, -1]              # k > n_tgt: the list ends, kdtree.hpp:66-67
    assert np.isinf(d2[0, 5:]).all() and np.allclose(d2[0, :2], [0.01, 0.81])
    assert (idx[1] == -1).all()                                      # NaN query: nothing compares less
    with pytest.raises(capi.IcpError):
        gpu_ctx.k_nearest(tgt, tgt, 0)
    with pytest.raises(capi.IcpError):
        gpu_ctx.k_nearest(np.zeros((0, 3)), tgt, 3)
    tree = icp.KDTree(tgt, ctx=gpu_ctx)
    assert tree.k_nearest([0.1, 0, 0], 3) == [0, 1, 2] and tree.nearest([4, 4, 4])[0] == 4
    assert tree.k_nearest([0.1, 0, 0], 9) == [0, 1, 2, 3, 4]


def test_ring_neighbourhood_normals_bit_exact(gpu_ctx, oracle):
    """Far-range LiDAR rings: every 20-neighbourhood is a nearly collinear arc, the covariance
    has two tiny, nearly equal eigenvalues -- where a Jacobi sweep and its GPU twin would part
    first if their operation orders differed.  Normals and neighbour lists must still be the
    oracle's bit for bit (the oracle itself is checked against numpy.linalg.eigh on this data in
    tests/test_oracle_residual_risk.py)."""
    from test_oracle_residual_risk import ring_cloud
    pts = ring_cloud(rings=8, per_ring=1800, seed=3)
    assert pts.shape[0] >= 8192                       # several splits per row
    got = gpu_ctx.estimate_normals(pts, 20)
    want = oracle.estimate_normals(pts, None, 20, nthreads=8)
    assert (got == want).all()
    idx, _ = gpu_ctx.k_nearest(pts, pts[::97], 20)
    tree = oracle.KDTree(pts)
    for r, i in enumerate(range(0, pts.shape[0], 97)):
        assert list(idx[r]) == list(tree.k_nearest(pts[i], 20))
    # a registration on this geometry: same iteration count and history as the oracle
    T = synth.make_transform((0.0, 0.0, 0.01), (0.2, -0.1, 0.0))
    src = (pts[::2] - T[:3, 3]) @ T[:3, :3]
    res, hist = gpu_ctx.align(src, pts, capi.Context.make_config(15, 1e-6, 1e-9))
    ref = oracle.icp_point_to_plane(src, pts, 15, 1e-6, 1e-9, nthreads=8)
    assert res.num_iterations == ref.num_iterations and bool(res.converged) == ref.converged
    np.testing.assert_allclose(hist, ref.error_history, rtol=0, atol=1e-9)


def test_c2_pair_at_kitti_size(gpu_ctx, oracle):
    """BASELINE.json configs[1] names ~20k points after the 0.5 m voxel filter (KITTI); the synthetic
    stand-in reaches that size at a 0.3 m voxel (17.9k -> 20.4k points).  Reference defaults
    (50 iterations, 1e-6, slam_node.cpp:134-136) against a live oracle run: same iteration count
    and flag, pose within the north_star's 1e-4 m / 1e-4 rad, history within 1e-9."""
    src, tgt, _ = synth.c2_lidar_pair(voxel=0.3)
    assert 17000 < src.shape[0] < 19000 and 19500 < tgt.shape[0] < 21500
    res, hist = gpu_ctx.align(src, tgt, capi.Context.make_config())
    ref = oracle.icp_point_to_plane(src, tgt, nthreads=8)
    assert res.num_iterations == ref.num_iterations and bool(res.converged) == ref.converged
    dt, dr = synth.pose_delta(np.array(res.transformation[:]).reshape(4, 4), ref.transformation)
    assert dt <= 1e-4 and dr <= 1e-4
    np.testing.assert_allclose(hist, ref.error_history, rtol=0, atol=1e-9)
    idx, d2 = gpu_ctx.nearest_batch(tgt, src)
    oidx, od2 = oracle.KDTree(tgt).nearest_batch(src, nthreads=8)
    assert (idx == oidx).all() and (d2 == od2).all()


def test_nonfinite_targets_are_never_neighbours(gpu_ctx, oracle):
    """A target with an Inf or NaN coordinate has no finite distance to anything, so no
    `dist_sq < best` (kdtree.hpp:125) ever selects it: every engine must return the nearest of
    the FINITE targets, bit for bit what the exhaustive oracle returns on the same array.
    (an Inf target used to turn its whole 2048-target split into NaN columns.)"""
    src, tgt, _ = synth.c3_uniform(12000, seed=61, perm_seed=62)
    tbad = tgt.copy()
    tbad[17, 0] = np.inf
    tbad[4000, 2] = -np.inf
    tbad[4001] = np.nan
    tbad[9000:9003, 1] = np.inf        # three in one slot
    tbad[11999, 0] = np.nan
    idx, d2 = gpu_ctx.nearest_batch(tbad, src)
    oidx, od2 = oracle.nearest_batch_brute(tbad, src)
    assert (idx == oidx).all() and (d2 == od2).all()
    bad_rows = np.flatnonzero(~np.isfinite(tbad).all(axis=1))
    assert not np.isin(idx, bad_rows).any()
    # a query far outside the cloud (squared distance beyond fp32): the bound must saturate, not turn NaN
    far = np.array([[1e25, -1e25, 3e24], [0.0, 0.0, 0.0]])
    fi, fd = gpu_ctx.nearest_batch(tbad, far)
    oi, od = oracle.nearest_batch_brute(tbad, far)
    assert (fi == oi).all() and (fd == od).all()


def test_profile_counters(gpu_ctx):
    src, tgt, _ = synth.c1_room_corner(2000)
    gpu_ctx.reset_profile()
    res, _ = gpu_ctx.align(src, tgt, capi.Context.make_config(5, 0.0, 0.0))
    p = gpu_ctx.get_profile()
    assert p["nn_launches"] == 6 and p["reduce_launches"] == 6 and p["normals_launches"] == 1
    assert p["nn_ms"] > 0 and p["total_ms"] >= p["loop_ms"] > 0
    assert p["nn_pairs"] == 6 * 2000 * 2000


@pytest.mark.parametrize("case", ["c3_small_20k", "lidar_pair", "far_start"])
def test_pruned_engine_skips_blocks_and_keeps_the_result(case, oracle):
    """ICPMI_SEARCH_MFMA_PRUNED culls (64-row group, target split) pairs with a bounding-box
    test against each group's exact distance to its previous neighbours (nn_culled.h).  The cull is
    conservative, so the correspondences -- hence error history, iteration count and pose
    -- are those of the all-pairs engine (sums are formed in Morton order of the source, so
    the last bits of the history may differ)."""
    if case == "c3_small_20k":
        src, tgt, _ = synth.c3_uniform(20000, seed=14, perm_seed=15)
        kw = dict(max_iterations=30, tolerance=0.0, min_error=0.0)
    elif case == "lidar_pair":
        src, tgt, _ = synth.c2_lidar_pair()
        kw = dict(max_iterations=50, tolerance=1e-6, min_error=1e-9)
    else:  # a start so far off that the first bounds prune nothing and most blocks miss the target box
        src, tgt, _ = synth.c3_uniform(12000, seed=3, perm_seed=4)
        kw = dict(max_iterations=8, tolerance=0.0, min_error=0.0,
                  initial_transform=synth.make_transform((0.3, -0.2, 0.4), (35.0, -20.0, 6.0)))
    cfg = capi.Context.make_config(**kw)
    full = capi.Context(device=0, search=capi.SEARCH_MFMA_BF16, profile=2)
    pruned = capi.Context(device=0, search=capi.SEARCH_MFMA_PRUNED, profile=2)
    try:
        a, ha = full.align(src, tgt, cfg)
        b, hb = pruned.align(src, tgt, cfg)
        p = pruned.get_profile()
        q = full.get_profile()
    finally:
        full.close()
        pruned.close()
    assert q["nn_pruned_blocks"] == 0 and q["nn_coarse_blocks"] > 0
    if case != "lidar_pair":   # (a run that converges queues a pass or two beyond its end, how many is a matter of timing)
        assert p["nn_coarse_blocks"] == q["nn_coarse_blocks"]
    if case == "c3_small_20k":
        assert p["nn_pruned_blocks"] > 0.3 * p["nn_coarse_blocks"], p
    elif case == "lidar_pair":   # ~3000-point frames: few, large blocks -- some culling only
        assert p["nn_pruned_blocks"] > 0, p
    assert a.num_iterations == b.num_iterations and bool(a.converged) == bool(b.converged)
    np.testing.assert_allclose(hb, ha, rtol=1e-12, atol=1e-15)
    dt, dr = synth.pose_delta(T_of(a), T_of(b))
    assert dt < 1e-10 and dr < 1e-10
    ref = oracle.icp_point_to_plane(src, tgt, **kw)
    check_against(b, hb, ref.transformation, ref.converged, ref.num_iterations, ref.error_history)


def _fuzz_cloud(rng, kind, n):
    """Clouds whose bounding boxes make culling hard: sheets, lines, clusters, far outliers."""
    if kind == "sheet":        # a thin, tilted plane: flat boxes overlapping many splits
        p = np.c_[rng.uniform(-40, 40, n), rng.uniform(-40, 40, n), rng.normal(0, 0.02, n)]
        R = synth.make_transform((0.3, 0.2, 0.1), (0, 0, 0))[:3, :3]
        return p @ R.T
    if kind == "lines":        # a few long segments: 1-D structure
        a = rng.uniform(-30, 30, (6, 3))
        b = rng.uniform(-30, 30, (6, 3))
        k = rng.integers(0, 6, n)
        t = rng.uniform(0, 1, (n, 1))
        return a[k] * (1 - t) + b[k] * t + rng.normal(0, 0.05, (n, 3))
    if kind == "clusters":     # dense blobs + sparse background
        c = rng.uniform(-50, 50, (25, 3))
        p = c[rng.integers(0, 25, n)] + rng.normal(0, 0.8, (n, 3))
        p[: n // 20] = rng.uniform(-60, 60, (n // 20, 3))
        return p
    if kind == "outliers":     # a compact cloud and a handful of points very far away
        p = rng.uniform(-10, 10, (n, 3))
        p[:8] = rng.uniform(-1, 1, (8, 3)) * 5.0e3
        return p
    raise ValueError(kind)


@pytest.mark.parametrize("kind,seed", [("sheet", 1), ("lines", 2), ("clusters", 3), ("outliers", 4),
                                       ("sheet", 5), ("clusters", 6)])
def test_engines_agree_on_awkward_geometry(kind, seed):
    """All-pairs vs pruned engine (and the fp64 engine on the first search) on clouds built to
    stress the bounding-box cull: identical correspondences -> identical history; sizes that are
    not multiples of the block (512) or the split (2048); a large initial misalignment."""
    rng = np.random.default_rng(seed)
    m = int(rng.integers(8200, 20000))
    n = int(rng.integers(3000, 15000))
    tgt = _fuzz_cloud(rng, kind, m)
    T = synth.make_transform(rng.normal(0, 0.03, 3), rng.normal(0, 0.4, 3))
    pick = rng.integers(0, m, n)
    src = (tgt[pick] + rng.normal(0, 0.01, (n, 3)) - T[:3, 3]) @ T[:3, :3]
    T0 = synth.make_transform(rng.normal(0, 0.2, 3), rng.normal(0, 3.0, 3)) if seed % 2 else np.eye(4)
    cfg = capi.Context.make_config(12, 0.0, 0.0, initial_transform=T0)
    out = {}
    for name, eng in (("all", capi.SEARCH_MFMA_BF16), ("pruned", capi.SEARCH_MFMA_PRUNED), ("f64", capi.SEARCH_EXACT_F64)):
        ctx = capi.Context(device=0, search=eng, profile=1)
        try:
            res, hist = ctx.align(src, tgt, cfg)
            moved = (src @ T0[:3, :3].T) + T0[:3, 3]
            out[name] = (T_of(res), hist, ctx.nearest_batch(tgt, moved)[0], ctx.estimate_normals(tgt, 20))
        finally:
            ctx.close()
    for other in ("pruned", "f64"):
        assert (out[other][2] == out["all"][2]).all()                       # first correspondences, bit for bit
        assert (out[other][3] == out["all"][3]).all()                       # normals, bit for bit
        assert len(out[other][1]) == len(out["all"][1])
        np.testing.assert_allclose(out[other][1], out["all"][1], rtol=1e-9, atol=1e-12)
        dt, dr = synth.pose_delta(out[other][0], out["all"][0])
        assert dt < 1e-8 and dr < 1e-8


def test_cpp_mirror_end_to_end(tmp_path, oracle):
    """include/icp_mi355x.hpp on the GPU from a plain C++17 program (tests/cpp/mirror_demo.cpp):
    NearestNeighborSearch, estimate_normals, solve_point_to_plane and ICP::align against the oracle."""
    import subprocess
    from lidar_slam_from_scratch_amd import build
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "mirror_demo"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "mirror_demo.cpp"), "-o", str(exe), build.LIB_PATH,
                           "-Wl,-rpath," + os.path.dirname(build.LIB_PATH), "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib",
                           "-lamdhip64"])
    src, tgt, _ = synth.c1_room_corner(3000)
    src.tofile(tmp_path / "s.f64")
    tgt.tofile(tmp_path / "t.f64")
    subprocess.check_call([str(exe), str(tmp_path / "s.f64"), str(tmp_path / "t.f64"), str(tmp_path / "o.f64")])
    o = np.fromfile(tmp_path / "o.f64")
    n, m = int(o[0]), int(o[1])
    assert (n, m) == (src.shape[0], tgt.shape[0])
    p = 2
    idx = o[p:p + n].astype(np.int64); p += n
    dist = o[p:p + n]; p += n
    nrm = o[p:p + 3 * m].reshape(m, 3); p += 3 * m
    Ts = o[p:p + 16].reshape(4, 4); p += 16
    Ti = o[p:p + 16].reshape(4, 4); p += 16
    conv, iters, ferr, hl = bool(o[p]), int(o[p + 1]), o[p + 2], int(o[p + 3]); p += 4
    hist = o[p:p + hl]
    oidx, od2 = oracle.KDTree(tgt).nearest_batch(src)
    assert (idx == oidx).all() and (dist == np.sqrt(od2)).all()
    onrm = oracle.estimate_normals(tgt, None, 20)
    assert (nrm == onrm).all()
    np.testing.assert_allclose(Ts, oracle.solve_point_to_plane(src, tgt[oidx], onrm[oidx]), atol=1e-12)
    ref = oracle.icp_point_to_plane(src, tgt)
    assert conv == ref.converged and iters == ref.num_iterations and ferr == hist[-1]
    dt, dr = synth.pose_delta(Ti, ref.transformation)
    assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD
    np.testing.assert_allclose(hist, ref.error_history, atol=HIST_TOL)


def test_widening_matches_golden(gpu_ctx):
    """Voxel filter and Scan Context on the GPU against tests/golden/widening.npz."""
    g = np.load(os.path.join(GOLD, "widening.npz"))
    raw = synth.lidar_frame(0, voxel=0, beams=32, azimuths=900)
    assert crc(raw) == g["raw_crc"], "generator drifted"
    vox = gpu_ctx.voxel_downsample(raw, float(g["voxel_size"]))
    assert vox.shape[0] == int(g["voxel_rows"]) and crc(vox) == g["voxel_crc"]
    desc = gpu_ctx.scan_context(vox)
    assert (desc != g["sc_desc"].reshape(desc.shape)).sum() <= 2      # atan2 last-ulp at sector edges
    d = gpu_ctx.scan_context_distances(g["sc_desc"], g["sc_desc_other"][None])
    assert d[0] == float(g["sc_distance"])


def test_context_churn_does_not_leak_device_memory():
    """Every icpmi_create is matched by an icpmi_destroy that returns all device memory
    (workspaces, events, stream, RCCL-free path): free memory after 40 create/align/destroy
    rounds, all engines, is what it was after the first."""
    import torch
    src, tgt, _ = synth.c3_uniform(12000, seed=8, perm_seed=9)
    cfg = capi.Context.make_config(3, 0.0, 0.0)

    def one(engine):
        ctx = capi.Context(device=0, search=engine, profile=2)
        ctx.align(src, tgt, cfg)
        ctx.voxel_downsample(tgt, 1.0)
        ctx.estimate_normals(tgt, 20)
        ctx.close()

    for e in (1, 2, 3):
        one(e)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for r in range(40):
        one(1 + r % 3)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 64 << 20, (free0, free1)


# ------------------------------------------------------------------ loop kernels: fused vs separate
def test_fused_finish_kernel_gives_the_separate_kernels_bits():
    """The single-GPU loop's k_finish_step_transform (default) against k_finish_step + k_transform
    (ICPMI_FUSE_FINISH=0): same sums in the same order, same step, same point update -- the pose, the
    history and the correspondences must agree bit for bit.  The knob is read once per process, so
    each leg is a child process; clouds of 300 (one workgroup, exact engine below 256 targets is not
    reached), 5,000 and 40,000 points (several workgroups each repeating the final sum)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = r'''
import sys, json, zlib
sys.path.insert(0, %r)
import numpy as np
from lidar_slam_from_scratch_amd import capi, synth
ctx = capi.Context(device=0)
out = {}
for n in (300, 5000, 40000):
    src, tgt, _ = synth.c3_uniform(n) if n > 5000 else synth.c1_room_corner(n)
    res, hist = ctx.align(src, tgt, capi.Context.make_config(max_iterations=12, tolerance=1e-7))
    out[str(n)] = {"T": [float.hex(v) for v in res.transformation[:]], "hist": [float.hex(v) for v in hist],
                   "iters": res.num_iterations, "converged": int(res.converged)}
print(json.dumps(out))
''' % root
    legs = {}
    for knob in ("0", "1"):
        env = dict(os.environ, ICPMI_FUSE_FINISH=knob)
        r = subprocess.run([sys.executable, "-c", child], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        legs[knob] = json.loads(r.stdout.strip().splitlines()[-1])
    assert legs["0"] == legs["1"]
    assert all(v["iters"] >= 2 for v in legs["1"].values())


def test_small_cloud_kernel_gives_the_general_path_bits():
    """icp_small.h: for targets of at most 16 splits (AUTO uses it up to 8; ICPMI_SMALL_MAX_SPLITS=16 here) the rows' work of an iteration (pose update, coarse
    pass, resolve, normal-equation terms) is ONE kernel whose intermediate never leaves LDS.  Against the
    general path (ICPMI_SMALL=0: k_nn_coarse + k_nn_resolve4 + k_finish_step_transform) the partial rows
    are formed from the same correspondences in the same order, so pose, history and counts must agree
    BIT FOR BIT: room corner at 300 / 5,000 points (1 / 3 splits), uniform clouds of 12k (6 splits) and
    20k -> 30k (15 splits), a LiDAR-like frame pair, a cloud full of exact ties (integer grid with
    duplicates), a source with a NaN row, and an initial transform."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = r'''
import sys, json
sys.path.insert(0, %r)
import numpy as np
from lidar_slam_from_scratch_amd import capi, synth
ctx = capi.Context(device=0, profile=1, search=capi.SEARCH_MFMA_BF16)   # (AUTO would take the culled engine on the 15-split target)
cases = {}
for n in (300, 5000):
    cases["corner%%d" %% n] = synth.c1_room_corner(n)[:2] + (None,)
cases["uniform12k"] = synth.c3_uniform(12000, seed=51, perm_seed=52)[:2] + (None,)
s, t, _ = synth.c3_uniform(30000, seed=61, perm_seed=62)
cases["uniform20k_30k"] = (s[:20000], t, None)
cases["lidar"] = (synth.lidar_frame(1), synth.lidar_frame(0), None)
rng = np.random.default_rng(5)
grid = rng.integers(-12, 12, (9000, 3)).astype(np.float64)
cases["ties"] = (grid[:4000] + 0.25, grid, None)
s, t, _ = synth.c1_room_corner(4000)
s = s.copy(); s[17] = np.nan
cases["nan_row"] = (s, t, None)
T0 = np.eye(4); T0[:3, 3] = (0.05, -0.02, 0.01)
cases["initial"] = synth.c1_room_corner(3000)[:2] + (T0,)
out = {}
for name, (src, tgt, init) in cases.items():
    cfg = capi.Context.make_config(max_iterations=12, tolerance=1e-7, initial_transform=init)
    res, hist = ctx.align(src, tgt, cfg)
    out[name] = {"T": [float.hex(v) for v in res.transformation[:]], "hist": [float.hex(v) for v in hist],
                 "iters": res.num_iterations, "converged": int(res.converged)}
out["small_launches"] = ctx.get_profile()["small_launches"]
print(json.dumps(out))
''' % root
    legs = {}
    for knob in ("0", "1"):
        env = dict(os.environ, ICPMI_SMALL=knob, ICPMI_SMALL_MAX_SPLITS="16")
        r = subprocess.run([sys.executable, "-c", child], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        legs[knob] = json.loads(r.stdout.strip().splitlines()[-1])
    assert legs["0"].pop("small_launches") == 0 and legs["1"].pop("small_launches") > 20
    assert legs["0"] == legs["1"]
    assert all(v["iters"] >= 1 for v in legs["1"].values())


def test_normal_lists_give_the_slot_minimum_form_bits():
    """knn_lists.h: normal estimation with the bound first and lists instead of slot minima, against round 2's form
    (ICPMI_KNN_LISTS=0), same GPU, child processes: both claim the exact k smallest by (distance, original index), so
    the normals must agree BIT FOR BIT also where the oracle's kd-tree would break ties its own way -- an integer grid
    with eight-fold duplicates (every neighbourhood full of exact ties, far more equidistant candidates than the
    128 a row keeps: tighten, go round again, or hand the row to the exact kernel), clusters, a line, a plane, mixed
    scales, a LiDAR-like frame, rows with NaN / infinite coordinates, one split and many; both MFMA engines."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = r'''
import sys, json, hashlib
sys.path.insert(0, %r); sys.path.insert(0, %r + "/scripts")
import numpy as np
from lidar_slam_from_scratch_amd import capi, synth
from fuzz_engines import cloud
rng = np.random.default_rng(3)
clouds = {}
g = rng.integers(-5, 6, (1500, 3)).astype(np.float64)
clouds["grid_dups_12k"] = np.repeat(g, 8, axis=0)[rng.permutation(12000)]
clouds["grid_70k"] = rng.integers(-20, 21, (70000, 3)).astype(np.float64) * 0.25
for kind in ("clusters", "line", "plane", "mixed"):
    clouds[kind + "_9k"] = cloud(rng, 9000, kind, 1.0, np.zeros(3))
    clouds[kind + "_40k"] = cloud(rng, 40000, kind, 10.0, np.array([1e3, -2e3, 50.0]))
clouds["lidar"] = synth.lidar_frame(0)
clouds["tiny_300"] = cloud(rng, 300, "uniform", 1.0, np.zeros(3))
bad = cloud(rng, 9000, "uniform", 1.0, np.zeros(3)); bad[5] = np.nan; bad[77, 1] = np.inf
clouds["nan_rows_9k"] = bad
bad = cloud(rng, 50000, "uniform", 1.0, np.zeros(3)); bad[5] = np.nan; bad[4077, 1] = -np.inf
clouds["nan_rows_50k"] = bad
out = {}
for engine in (capi.SEARCH_MFMA_BF16, capi.SEARCH_MFMA_PRUNED):
    ctx = capi.Context(device=0, search=engine)
    for name, pts in clouds.items():
        for k in (20, 5):
            nrm = ctx.estimate_normals(pts, k)
            out["%%d %%s k%%d" %% (engine, name, k)] = hashlib.sha256(np.ascontiguousarray(nrm).tobytes()).hexdigest()
    ctx.close()
print(json.dumps(out))
''' % (root, root)
    legs = {}
    for knob in ("0", "1"):
        env = dict(os.environ, ICPMI_KNN_LISTS=knob)
        r = subprocess.run([sys.executable, "-c", child], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        legs[knob] = json.loads(r.stdout.strip().splitlines()[-1])
    differing = [k for k in legs["0"] if legs["0"][k] != legs["1"][k]]
    assert not differing, differing


def test_bounded_pass_gives_the_unbounded_pass_bits():
    """nn_bounded.h: every pass of the ICP loop searches behind a bound per row -- the exact distance to the row's
    previous match, in the first pass to the nearest sorted target around its place in the target's Morton order
    (k_nn_prebound1, round 4) -- the coarse pass lists the slots under each row's bound instead of keeping its minima,
    the resolve scans the listed slots against the bound's target as incumbent, no certificate.  Against
    ICPMI_NN_BOUNDED=0 on the all-pairs engine (every pass k_nn_coarse<0> + k_nn_resolve / k_nn_resolve4) the
    correspondences, hence the partial rows, pose, history and counts must agree BIT FOR BIT -- the all-pairs engine's
    bounded passes AND the culled engine's (nn_culled.h: the same passes on the (64-row group, split) pairs within
    reach; what AUTO runs on these targets) -- for both resolve layouts (16 rows per wave above 32,768 rows, one per quarter-wave below) and both coarse
    unit sizes: uniform clouds, a raw
    LiDAR-like frame pair, a cloud full of exact ties, a source with NaN / infinite rows, a target with NaN / infinite points
    (their keys sort to one corner: the first pass's window there holds no candidate), a start so far off that the
    bounds span the whole target (every row takes the exhaustive search behind the split boxes), and two ranks."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = r'''
import sys, json
sys.path.insert(0, %r)
import numpy as np
from lidar_slam_from_scratch_amd import capi, synth, dist as icpdist
cases = {}
s, t, _ = synth.c3_uniform(60000, seed=71, perm_seed=72)
cases["uniform60k"] = (s, t, None, 8)
cases["uniform40k_60k"] = (s[:40000], t, None, 6)
cases["shard12k_60k"] = (s[:12500], t, None, 6)          # quarter-wave resolve (k_nn_resolve4_bounded<8>)
cases["few_units_5k_40k"] = (s[:5000], t[:40000], None, 6)  # 256-query coarse units
cases["lidar_raw"] = (synth.lidar_frame(1, voxel=0), synth.lidar_frame(0, voxel=0), None, 6)
rng = np.random.default_rng(5)
grid = rng.integers(-20, 20, (70000, 3)).astype(np.float64)
cases["ties"] = (grid[:40000] + 0.25, grid, None, 5)
s2 = s.copy(); s2[17] = np.nan; s2[40001, 1] = np.inf
cases["nan_rows"] = (s2, t, None, 5)
t2 = t.copy(); t2[11] = np.nan; t2[30001, 2] = np.inf; t2[59999, 0] = -np.inf   # targets nobody may be matched with (kdtree.hpp:125)
cases["nan_targets"] = (s[:50000], t2, None, 4)
T0 = synth.make_transform(np.array([0.3, -0.2, 0.25]), np.array([40.0, -25.0, 10.0]))
cases["far_start"] = (s[:36000], t, T0, 4)
out = {}
engine = int(sys.argv[1])
ctx = capi.Context(device=0, profile=2, search=engine)   # (2: the bounded passes report their statistics too)
for name, (src, tgt, init, iters) in cases.items():
    cfg = capi.Context.make_config(max_iterations=iters, tolerance=0.0, min_error=0.0, initial_transform=init)
    ctx.reset_profile()
    res, hist = ctx.align(src, tgt, cfg)
    p = ctx.get_profile()
    out[name] = {"T": [float.hex(v) for v in res.transformation[:]], "hist": [float.hex(v) for v in hist],
                 "iters": res.num_iterations, "bounded": int(p["bounded_launches"]), "exhaustive": int(p["nn_fallback_queries"])}
ctx.close()
# two ranks (threads of this process, host-callback exchange), 45k rows each
group = icpdist.LocalGroup(2)
s, t, _ = synth.c3_uniform(90000, seed=81, perm_seed=82)
def body(rank):
    lo, hi = icpdist.shard_bounds(s.shape[0], 2, rank)
    c = capi.Context(device=0, search=engine)
    group.attach(c, rank)
    res, hist = c.align(s[lo:hi], t, capi.Context.make_config(max_iterations=4, tolerance=0.0, min_error=0.0))
    c.comm_finalize(); c.close()
    return [float.hex(v) for v in res.transformation[:]], [float.hex(v) for v in hist]
r = group.run(body)
assert r[0] == r[1]
out["two_ranks"] = {"T": r[0][0], "hist": r[0][1]}
print(json.dumps(out))
''' % root
    legs = {}
    for knob, engine in (("0", capi.SEARCH_MFMA_BF16), ("1", capi.SEARCH_MFMA_BF16), ("culled", capi.SEARCH_AUTO)):
        env = dict(os.environ, ICPMI_NN_BOUNDED="0" if knob == "0" else "1")
        r = subprocess.run([sys.executable, "-c", child, str(engine)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        legs[knob] = json.loads(r.stdout.strip().splitlines()[-1])
    for knob in ("1", "culled"):
        for name, v in legs[knob].items():
            if name == "two_ranks":
                continue
            assert v.pop("bounded") == v["iters"] + 1, (knob, name)   # every pass, the first and the post-loop pass included
            ex1 = v.pop("exhaustive")
            if name == "far_start":
                assert ex1 > 30000, (knob, ex1)                        # the exhaustive path did run
    for name, v in legs["0"].items():
        if name != "two_ranks":
            assert v.pop("bounded") == 0
            v.pop("exhaustive")
    assert legs["0"] == legs["1"]
    assert legs["0"] == legs["culled"]


def test_align_batch_is_the_sequential_calls(gpu_ctx):
    """icpmi_align_batch: the up-to-three verifications of one LoopClosureDetector::detect
    (loop_closure.hpp:94-123) side by side, each on a stream and workspace of its own.  Every result must
    be BIT-identical to the same registration made alone; a problem that fails (empty target) reports
    its own status and the call returns it; the context stays usable."""
    q = synth.lidar_frame(1, beams=32, azimuths=900)
    tgts = [synth.lidar_frame(f, beams=32, azimuths=900) for f in (0, 2, 3)]
    cfg = capi.Context.make_config(max_iterations=30, tolerance=1e-6)     # loop_closure.hpp:105-107
    alone = [gpu_ctx.align(q, t, cfg) for t in tgts]
    for rep in range(2):                                                  # (second time: helper contexts already there)
        both = gpu_ctx.align_batch([q] * 3, tgts, cfg)
        for (ra, ha), (rb, hb) in zip(alone, both):
            assert tuple(ra.transformation) == tuple(rb.transformation) and (ha == hb).all()
            assert (ra.converged, ra.num_iterations, ra.final_error) == (rb.converged, rb.num_iterations, rb.final_error)
    # different sources and per-problem configs; one problem
    s2 = synth.c1_room_corner(3000)
    cfgs = [capi.Context.make_config(max_iterations=5, tolerance=0.0, min_error=0.0), cfg]
    two = gpu_ctx.align_batch([s2[0], q], [s2[1], tgts[0]], cfgs)
    r0, h0 = gpu_ctx.align(s2[0], s2[1], cfgs[0])
    assert tuple(two[0][0].transformation) == tuple(r0.transformation) and (two[0][1] == h0).all() and two[0][0].num_iterations == 5
    assert tuple(two[1][0].transformation) == tuple(alone[0][0].transformation)
    one = gpu_ctx.align_batch([q], [tgts[1]], cfg)
    assert tuple(one[0][0].transformation) == tuple(alone[1][0].transformation)
    with pytest.raises(capi.IcpError) as e:
        gpu_ctx.align_batch([q, q], [tgts[0], np.zeros((0, 3))], cfg)
    assert e.value.code == capi.ERR_EMPTY_TARGET
    with pytest.raises(capi.IcpError):
        gpu_ctx.align_batch([q] * 9, [tgts[0]] * 9, cfg)                  # more than ICPMI_MAX_BATCH
    again = gpu_ctx.align(q, tgts[2], cfg)
    assert tuple(again[0].transformation) == tuple(alone[2][0].transformation)


def test_wave_step_gives_the_one_lane_step_bits(tmp_path):
    """The pose update of every iteration (6x6 LDLT solve, Rodrigues, 4x4 product: icp.hpp:120-143, 229) is done by a
    whole wave since round 4 (device_math.h ldlt6_solve_wave: the pivot order from the input diagonal, one lane
    permutation, a column per step; mul44_wave).  The one-lane form it replaced restates the oracle's operation order
    statement by statement and stays in the header: scripts/micro/step_clocks.hip runs both on 60,000 systems -- well
    conditioned, rank-deficient (one plane, normals in a plane, no lever arm), tie-heavy, small-integer indefinite, all
    zero, NaN -- and compares solution, step transform and accumulated transform bit for bit (two NaNs are equal); it
    also bounds sincos_step against the device library's sin / cos (<= 2 ulp; measured 1.00 / 0.71)."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = tmp_path / "step_clocks"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-I",
                           os.path.join(ROOT, "lidar_slam_from_scratch_amd", "csrc"),
                           os.path.join(ROOT, "scripts", "micro", "step_clocks.hip"), "-o", str(exe)], stderr=subprocess.DEVNULL)
    out = subprocess.run([str(exe), "64", "2000", "60000"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:]
    assert "60000 systems: 0 differ" in out.stdout, out.stdout[-2000:]
