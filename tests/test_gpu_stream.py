"""SURVEY section 8f rows N3 / N4 and configs C4 / C5 on the GPU: scans from disk to HBM as
float32, the resident-frame odometry call of the C ABI, and one 125k-row shard of the 1M -> 1M
job.  Parity targets: the host loader, the oracle-driven loop, cKDTree."""
import os
import sys

import numpy as np
import pytest
import torch

from lidar_slam_from_scratch_amd import capi, odometry, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(device=0)
    yield c
    c.close()


def test_bin_goes_to_the_device_as_float32(tmp_path, ctx):
    """icpmi_load_cloud_device: KITTI .bin (file_utils.cpp:115-141) widened on the device must be
    the host loader's array bit for bit; PLY takes the host parser."""
    rng = np.random.default_rng(1)
    rec = rng.normal(size=(5003, 4)).astype(np.float32) * 40
    rec[7, 0] = np.float32(1e-30)
    (tmp_path / "000000.bin").write_bytes(rec.tobytes() + b"\x01\x02\x03")          # trailing partial record ignored
    path = str(tmp_path / "000000.bin")
    want = capi.load_cloud(path)
    n = ctx.load_cloud_device_rows(path)
    assert n == 5003
    d = torch.zeros((n + 5, 3), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    assert ctx.load_cloud_device(path, d.data_ptr(), d.shape[0]) == n
    got = d.cpu().numpy()
    assert (got[:n] == want).all() and (got[n:] == 0).all()
    assert (want == rec[:, :3].astype(np.float64)).all()
    with pytest.raises(capi.IcpError):
        ctx.load_cloud_device(path, d.data_ptr(), 10)                               # capacity
    with pytest.raises(capi.IcpError):
        ctx.load_cloud_device(str(tmp_path / "nope.bin"), d.data_ptr(), d.shape[0])
    # generic float32 records (stride 5)
    rec5 = rng.normal(size=(100, 5)).astype(np.float32)
    ctx.upload_points_f32(rec5, d.data_ptr())
    assert (d.cpu().numpy()[:100] == rec5[:, :3].astype(np.float64)).all()
    # PLY (ASCII) through the same entry point
    ply = tmp_path / "c.ply"
    ply.write_text("ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\n"
                   "end_header\n1 2 3\n4.5 5.5 6.5\n-7 8 9e-3\n")
    assert ctx.load_cloud_device(str(ply), d.data_ptr(), d.shape[0]) == 3
    assert (d.cpu().numpy()[:3] == capi.load_cloud(str(ply))).all()


def test_normals_of_a_row_slice(ctx):
    _, tgt, _ = synth.c3_uniform(12000, seed=81, perm_seed=82)
    full = ctx.estimate_normals(tgt, 20)
    assert (ctx.estimate_normals_rows(tgt, 20, 3000, 7000) == full[3000:7000]).all()
    assert (ctx.estimate_normals_rows(tgt, 20, 11990, 12000) == full[11990:]).all()
    assert ctx.estimate_normals_rows(tgt, 20, 5, 5).shape == (0, 3)
    with pytest.raises(capi.IcpError):
        ctx.estimate_normals_rows(tgt, 20, 10, 12001)


def test_stream_push_is_process_frame(tmp_path, oracle):
    """30 frames of the synthetic drive (SURVEY 8d C5 stand-in) as KITTI .bin files: discover_frames,
    file -> HBM, icpmi_stream_push per frame.  Iteration counts, gates and poses must be those of
    the reference loop (slam_node.cpp:118-157) driven by the oracle on the host-loaded, host-filtered
    clouds; ATE against the known trajectory equal to the oracle's.  Then the guards."""
    import run_sequence
    truth = run_sequence.write_synthetic_drive(str(tmp_path), 0, 30, beams=32, azimuths=900)
    frames = capi.discover_frames(str(tmp_path))
    assert [k for k, _ in frames] == list(range(30))
    paths = [p for _, p in frames]
    ctx = capi.Context(device=0)
    tr = odometry.run_odometry_stream(paths, ctx, voxel=0.5, min_points=1000)
    clouds = [oracle.voxel_downsample(capi.load_cloud(p), 0.5) for p in paths]
    rf = odometry.run_odometry(clouds, lambda s, t, mi, tol: oracle.icp_point_to_plane(s, t, mi, tol, 1e-9, nthreads=8))
    assert tr.iterations == rf.iterations and tr.gated == rf.gated and tr.converged == rf.converged
    assert len(tr.poses) == 30
    for a, b in zip(tr.poses, rf.poses):
        dt, dr = synth.pose_delta(a, b)
        assert dt <= 1e-4 and dr <= 1e-4                       # north_star tolerance (measured ~1e-12)
    np.testing.assert_allclose(tr.final_errors, rf.final_errors, rtol=0, atol=1e-9)
    ate_g, ate_c = odometry.absolute_trajectory_error(tr, truth), odometry.absolute_trajectory_error(rf, truth)
    assert abs(ate_g - ate_c) < 1e-6      # (the drive is a corridor: both loops drift alike along it)
    # the same drive through the torch-side loop of round 1 (clouds uploaded by the caller)
    raw = [capi.load_cloud(p) for p in paths[:8]]
    td = odometry.run_odometry_device(raw, ctx, voxel=0.5)
    assert td.iterations == tr.iterations[:7]
    # the device-pointer form with the scan loaded by icpmi_load_cloud_device (what round 2's first harness did)
    ctx.stream_reset()
    its = []
    dbuf = torch.empty((max(r.shape[0] for r in raw) + 8, 3), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    for p in paths[:8]:
        n = ctx.load_cloud_device(p, dbuf.data_ptr(), dbuf.shape[0])
        res, _, info = ctx.stream_push(dbuf.data_ptr(), n, 0.5, 1000, capi.Context.make_config())
        if info.status == capi.STREAM_REGISTERED:
            its.append(res.num_iterations)
    assert its == tr.iterations[:7]
    # ... and through the host-pointer form the Eigen adapter's OdometryStream uses
    ctx.stream_reset()
    its = []
    for r in raw[:8]:
        res, _, info = ctx.stream_push_host(r, 0.5, 1000, capi.Context.make_config())
        if info.status == capi.STREAM_REGISTERED:
            its.append(res.num_iterations)
    assert its == tr.iterations[:7]
    # guards: a frame below min_points repeats the pose and still becomes the next target (slam_node.cpp:125-130)
    ctx.stream_reset()
    cfg = capi.Context.make_config()
    d = [torch.from_numpy(r).cuda() for r in raw[:3]]
    few = np.ascontiguousarray(raw[1][np.random.default_rng(4).choice(raw[1].shape[0], 700, replace=False)])
    tiny = torch.from_numpy(few).cuda()             # spread over the whole scan, but under min_points
    torch.cuda.synchronize()
    _, _, i0 = ctx.stream_push(d[0].data_ptr(), d[0].shape[0], 0.5, 1000, cfg)
    r1, h1, i1 = ctx.stream_push(tiny.data_ptr(), tiny.shape[0], 0.5, 1000, cfg)
    r2, h2, i2 = ctx.stream_push(d[2].data_ptr(), d[2].shape[0], 0.5, 1000, cfg)
    assert (i0.status, i1.status, i2.status) == (capi.STREAM_FIRST_FRAME, capi.STREAM_TOO_FEW_POINTS, capi.STREAM_REGISTERED)
    assert i1.n_filtered < 1000 and i2.n_target == i1.n_filtered and not r1.converged and len(h1) == 0
    assert list(r1.transformation) == list(np.eye(4).reshape(16))
    # frames as binary PLY (what the node really loads, slam_node.cpp:121): the host parser feeds the same stream
    ctx.stream_reset()
    its = []
    for k in range(3):
        rec = np.zeros((raw[k].shape[0], 4), dtype=np.float32)
        rec[:, :3] = raw[k]
        ply = tmp_path / ("%06d.ply" % k)
        with open(ply, "wb") as f:
            f.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\n"
                     "property float z\nproperty float intensity\nend_header\n" % rec.shape[0]).encode())
            f.write(rec.tobytes())
        res, _, info = ctx.stream_push_file(str(ply), 0.5, 1000, cfg)
        if info.status == capi.STREAM_REGISTERED:
            its.append(res.num_iterations)
    assert its == tr.iterations[:2]
    with pytest.raises(capi.IcpError):
        ctx.stream_push_file(str(tmp_path / "missing.bin"), 0.5, 1000, cfg)
    want = oracle.icp_point_to_plane(oracle.voxel_downsample(raw[2], 0.5), oracle.voxel_downsample(few, 0.5))
    assert r2.num_iterations == want.num_iterations and abs(r2.final_error - want.final_error) < 1e-9
    ctx.close()


def test_c4_shard_properties(oracle):
    """BASELINE.json configs[3]: 1M -> 1M sharded 8 ways; what ONE rank does -- a 125k-row shard of the
    source against the whole 1M target (489 splits, 489 MB of coarse minima, the >32k-query resolve) --
    on one GPU, through properties that need no 1M x 125k brute force on the host: index range, the
    returned squared distance recomputed exactly, cKDTree on a 20k sample, normals of a 20k-row
    slice against the oracle, and two iterations of the registration against the oracle loop."""
    from scipy.spatial import cKDTree
    src, tgt, _ = synth.c4_uniform()
    assert tgt.shape[0] == 1_000_000
    shard = np.ascontiguousarray(src[:125_000])
    ctx = capi.Context(device=0)
    idx, d2 = ctx.nearest_batch(tgt, shard)
    assert idx.min() >= 0 and idx.max() < tgt.shape[0]
    diff = tgt[idx] - shard
    assert (d2 == (diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1]) + diff[:, 2] * diff[:, 2]).all()
    tree = cKDTree(tgt)
    sel = np.random.default_rng(0).choice(shard.shape[0], 20_000, replace=False)
    dd, ii = tree.query(shard[sel], workers=-1)
    assert (idx[sel] == ii).all()
    nrm = ctx.estimate_normals_rows(tgt, 20, 500_000, 520_000)
    # bit for bit against the oracle's normals of the same rows, like every other normals test
    want = oracle.estimate_normals_rows(tgt, 500_000, 520_000, None, 20, nthreads=os.cpu_count() or 1)
    assert (nrm == want).all()
    res, hist = ctx.align(shard, tgt, capi.Context.make_config(2, 0.0, 0.0))
    ref = oracle.icp_point_to_plane(shard, tgt, 2, 0.0, 0.0, faithful=False, nthreads=16)
    assert res.num_iterations == ref.num_iterations == 2
    np.testing.assert_allclose(hist, ref.error_history, rtol=0, atol=1e-9)
    dt, dr = synth.pose_delta(np.array(res.transformation[:]).reshape(4, 4), ref.transformation)
    assert dt <= 1e-9 and dr <= 1e-9
    ctx.close()


def test_prefetched_frames_give_the_same_stream(tmp_path):
    """icpmi_stream_prefetch_file: the next frame's file read by the context's worker thread while the current
    frame runs.  Same iteration counts, poses and errors as without; a prefetched path that is not the one
    pushed next, a missing file and a PLY (no-op) must all leave the stream intact."""
    import run_sequence
    run_sequence.write_synthetic_drive(str(tmp_path), 0, 12, beams=32, azimuths=900)
    paths = [p for _, p in capi.discover_frames(str(tmp_path))]
    ctx = capi.Context(device=0)
    a = odometry.run_odometry_stream(paths, ctx, prefetch=False)
    b = odometry.run_odometry_stream(paths, ctx, prefetch=True)
    assert a.iterations == b.iterations and a.gated == b.gated
    assert all((x == y).all() for x, y in zip(a.poses, b.poses)) and a.final_errors == b.final_errors
    cfg = capi.Context.make_config()
    ctx.stream_reset()
    ctx.stream_prefetch_file(paths[5])                       # not the one pushed next: the push reads its own file
    _, _, i0 = ctx.stream_push_file(paths[0], 0.5, 1000, cfg)
    r1, _, i1 = ctx.stream_push_file(paths[1], 0.5, 1000, cfg)
    assert (i0.status, i1.status) == (capi.STREAM_FIRST_FRAME, capi.STREAM_REGISTERED) and r1.num_iterations == a.iterations[0]
    ctx.stream_prefetch_file(str(tmp_path / "missing.bin"))  # cannot be read: reported by the push of that path only
    r2, _, _ = ctx.stream_push_file(paths[2], 0.5, 1000, cfg)
    assert r2.num_iterations == a.iterations[1]
    with pytest.raises(capi.IcpError):
        ctx.stream_push_file(str(tmp_path / "missing.bin"), 0.5, 1000, cfg)
    ctx.stream_prefetch_file(str(tmp_path / "x.ply"))        # PLY: nothing to do ahead
    ctx.stream_prefetch_file(paths[3])
    ctx.stream_prefetch_file(paths[3])                       # twice: the second waits for the first
    r3, _, _ = ctx.stream_push_file(paths[3], 0.5, 1000, cfg)
    assert r3.num_iterations == a.iterations[2]
    # two files ahead, then three (the oldest gives way and is read by its own push), in stream order throughout
    ctx.stream_prefetch_file(paths[4])
    ctx.stream_prefetch_file(paths[5])
    got = [ctx.stream_push_file(paths[k], 0.5, 1000, cfg)[0].num_iterations for k in (4, 5)]
    ctx.stream_prefetch_file(paths[6])
    ctx.stream_prefetch_file(paths[7])
    ctx.stream_prefetch_file(paths[8])
    got += [ctx.stream_push_file(paths[k], 0.5, 1000, cfg)[0].num_iterations for k in (6, 7, 8)]
    # a push with another voxel size than the worker filtered with: the raw points in the slot are filtered by the push
    ctx.stream_prefetch_file(paths[9])
    r9, _, i9 = ctx.stream_push_file(paths[9], 0.25, 1000, cfg)
    assert got == a.iterations[3:8] and i9.status == capi.STREAM_REGISTERED and i9.n_filtered > 1.5 * i1.n_filtered
    ctx.stream_prefetch_file(paths[10])                      # left pending: close() must stop the worker
    ctx.close()


def test_prefetched_files_the_worker_cannot_filter(tmp_path):
    """The worker queues the filter of a prefetched file without looking at its outcome; the push of that file
    does.  A scan whose grid cannot be keyed (a stray return three thousand kilometres away): the push then filters
    the raw points itself and reports the same error as without the worker.  A file with trailing bytes (not a whole record) and a three-point file go
    through like in the synchronous path.  The stream stays usable throughout."""
    import run_sequence
    run_sequence.write_synthetic_drive(str(tmp_path), 0, 4, beams=32, azimuths=900)
    paths = [p for _, p in capi.discover_frames(str(tmp_path))]
    cfg = capi.Context.make_config()
    rec = np.fromfile(paths[2], dtype=np.float32).reshape(-1, 4)
    bad = rec.copy(); bad[17, 1] = 3.0e6                      # 6e6 cells of 0.5 m on one axis: more than the 2^21 a key holds
    bad_path = str(tmp_path / "900001.bin"); bad.tofile(bad_path)
    odd_path = str(tmp_path / "900002.bin")
    with open(odd_path, "wb") as f:
        f.write(rec.tobytes() + b"\x01\x02\x03\x04\x05")       # file_utils.cpp:127: size / 16 records, the rest ignored
    tiny_path = str(tmp_path / "900003.bin"); rec[:3].tofile(tiny_path)

    def run(prefetch):
        ctx = capi.Context(device=0)
        out = []
        for p in (paths[0], paths[1], bad_path, paths[2], odd_path, tiny_path, paths[3]):
            if prefetch:
                ctx.stream_prefetch_file(p)
            try:
                r, _, info = ctx.stream_push_file(p, 0.5, 1000, cfg)
                out.append((info.status, info.n_filtered, r.num_iterations, tuple(r.transformation)))
            except capi.IcpError as e:
                out.append(("error", e.code, str(e)))
        ctx.close()
        return out

    a, b = run(False), run(True)
    assert a == b
    assert a[2][0] == "error" and a[2][1] == capi.ERR_ARG and "voxel" in a[2][2]
    assert a[3][0] == capi.STREAM_REGISTERED and a[4][0] == capi.STREAM_REGISTERED and a[4][1] == a[3][1]
    assert a[5][0] == capi.STREAM_TOO_FEW_POINTS     # (and the frame after it registers against those three points: a == b covers it)


def test_early_target_preparation_survives_foreign_calls(tmp_path, oracle):
    """A push queues the search structure and normals of the scan it has just filtered (the next push's target)
    behind its own result.  Calls in between that use the same buffers for other clouds -- a search, normals, a
    registration, a voxel filter, a map update -- must make the next push prepare its target again; the stream's
    results stay those of the undisturbed stream, and a context with stage timers (no early preparation) agrees."""
    import run_sequence
    run_sequence.write_synthetic_drive(str(tmp_path), 0, 8, beams=32, azimuths=900)
    paths = [p for _, p in capi.discover_frames(str(tmp_path))]
    ctx = capi.Context(device=0)
    plain = odometry.run_odometry_stream(paths, ctx, prefetch=False)
    timed = capi.Context(device=0, profile=2)
    with_timers = odometry.run_odometry_stream(paths, timed, prefetch=False)
    assert with_timers.iterations == plain.iterations and all((a == b).all() for a, b in zip(with_timers.poses, plain.poses))
    timed.close()
    rng = np.random.default_rng(2)
    other_a, other_b = rng.uniform(-20, 20, (3000, 3)), rng.uniform(-20, 20, (2500, 3))
    cfg = capi.Context.make_config()
    ctx.stream_reset()
    its, Ts = [], []
    for k, p in enumerate(paths):
        res, _, info = ctx.stream_push_file(p, 0.5, 1000, cfg)
        if info.status == capi.STREAM_REGISTERED:
            its.append(res.num_iterations)
            Ts.append(np.array(res.transformation[:]))
        # something else happens on the context between two frames
        if k % 4 == 0:
            idx, _ = ctx.nearest_batch(other_a, other_b)
            assert (idx == oracle.KDTree(other_a).nearest_batch(other_b)[0]).all()
        elif k % 4 == 1:
            ctx.estimate_normals(other_a, 20)
        elif k % 4 == 2:
            ctx.align(other_b, other_a, capi.Context.make_config(max_iterations=3))
        else:
            ctx.voxel_downsample(other_a, 0.5)
            ctx.stream_map_update(np.eye(4))
    assert its == plain.iterations
    for T, d in zip(Ts, plain.deltas):
        if not (d == np.eye(4)).all():          # (gated frames carry the identity in the track)
            assert (T.reshape(4, 4) == d).all()
    ctx.close()


def test_c5_stream_at_stream_length(tmp_path, oracle):
    """BASELINE.json configs[4] at the length SURVEY 8(d) gives its stand-in (KITTI is absent, F4): the
    200-frame synthetic drive as KITTI .bin files (64 beams x 1800 azimuths, ~115k points per raw scan,
    ~7k after the 0.5 m filter), file -> pose through icpmi_stream_prefetch_file + icpmi_stream_push_file,
    against the reference loop (slam_node.cpp:118-157) driven by the oracle on host-loaded, host-filtered
    clouds: iteration counts, convergence flags and gates of all 199 registrations equal, poses within the
    north_star's 1e-4 m / 1e-4 rad (measured ~1e-13), ATE against the known trajectory equal."""
    import run_sequence
    nth = os.cpu_count() or 1
    truth = run_sequence.write_synthetic_drive(str(tmp_path), 0, 200, workers=min(16, nth))
    paths = [p for _, p in capi.discover_frames(str(tmp_path))]
    assert len(paths) == 200
    ctx = capi.Context(device=0)
    tr = odometry.run_odometry_stream(paths, ctx, voxel=0.5, min_points=1000, prefetch=True)
    ctx.close()
    clouds = (oracle.voxel_downsample(capi.load_cloud(p), 0.5) for p in paths)
    rf = odometry.run_odometry(clouds, lambda s, t, mi, tol: oracle.icp_point_to_plane(s, t, mi, tol, 1e-9, nthreads=nth))
    assert len(tr.poses) == len(rf.poses) == 200
    assert tr.iterations == rf.iterations and tr.converged == rf.converged and tr.gated == rf.gated
    worst_t = worst_r = 0.0
    for a, b in zip(tr.poses, rf.poses):
        dt, dr = synth.pose_delta(a, b)
        worst_t, worst_r = max(worst_t, dt), max(worst_r, dr)
    assert worst_t <= 1e-4 and worst_r <= 1e-4, (worst_t, worst_r)
    np.testing.assert_allclose(tr.final_errors, rf.final_errors, rtol=0, atol=1e-9)
    ate_g, ate_c = odometry.absolute_trajectory_error(tr, truth), odometry.absolute_trajectory_error(rf, truth)
    assert abs(ate_g - ate_c) < 1e-6
    print("C5 stand-in, 200 frames: %d iterations, %d gated, worst pose delta %.2e m / %.2e rad, ATE %.7f / %.7f m"
          % (sum(tr.iterations), sum(tr.gated), worst_t, worst_r, ate_g, ate_c))


def test_failed_push_leaves_the_prefetched_frame(tmp_path):
    """a push of a prefetched-and-filtered file that fails (here: arguments the call rejects) must not
    consume the frame -- its filtered copy is not swapped away and the slot stays taken -- so that the same
    push, repeated with good arguments, gives what an undisturbed stream gives."""
    import run_sequence
    run_sequence.write_synthetic_drive(str(tmp_path), 0, 4, beams=32, azimuths=900)
    paths = [p for _, p in capi.discover_frames(str(tmp_path))]
    ctx = capi.Context(device=0)
    want = odometry.run_odometry_stream(paths, ctx, prefetch=True)
    cfg = capi.Context.make_config()
    bad = capi.Context.make_config(max_iterations=-1)
    ctx.stream_reset()
    its = []
    for k, p in enumerate(paths):
        if k + 1 < len(paths):
            ctx.stream_prefetch_file(paths[k + 1])
        if k == 2:
            with pytest.raises(capi.IcpError) as e:
                ctx.stream_push_file(p, 0.5, 1000, bad)
            assert e.value.code == capi.ERR_ARG
        res, _, info = ctx.stream_push_file(p, 0.5, 1000, cfg)
        if info.status == capi.STREAM_REGISTERED:
            its.append((res.num_iterations, tuple(res.transformation)))
    assert [i for i, _ in its] == want.iterations
    for (_, T), d in zip(its, want.deltas):
        if not (d == np.eye(4)).all():
            assert (np.array(T).reshape(4, 4) == d).all()
    ctx.close()
