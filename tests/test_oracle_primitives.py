"""KAT-2: the oracle's primitives against independent NumPy/SciPy implementations
(SURVEY.md section 8c).  The reference has no tests of its own (parity unpinned)."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st
from scipy.spatial import cKDTree
from scipy.spatial.transform import Rotation

from lidar_slam_from_scratch_amd import synth


def test_nearest_matches_ckdtree(oracle):
    src, tgt, _ = synth.c1_room_corner()
    idx, d2 = oracle.KDTree(tgt).nearest_batch(src)
    dd, ii = cKDTree(tgt).query(src)
    assert (idx == ii).all()
    np.testing.assert_allclose(np.sqrt(d2), dd, rtol=1e-14, atol=0)


def test_kdtree_equals_bruteforce(oracle):
    src, tgt, _ = synth.c3_uniform(3000, seed=21, perm_seed=22)
    idx, d2 = oracle.KDTree(tgt).nearest_batch(src)
    bi, bd = oracle.nearest_batch_brute(tgt, src)
    assert (idx == bi).all() and (d2 == bd).all()


def test_threaded_batch_equals_serial(oracle):
    src, tgt, _ = synth.c3_uniform(4000, seed=23, perm_seed=24)
    t = oracle.KDTree(tgt)
    a, ad = t.nearest_batch(src, nthreads=1)
    b, bd = t.nearest_batch(src, nthreads=4)
    assert (a == b).all() and (ad == bd).all()


@settings(max_examples=25, deadline=None)
@given(st.integers(1, 300), st.integers(1, 40), st.integers(0, 2**31 - 1))
def test_kdtree_bruteforce_property(n_tgt, n_qry, seed):
    from oracle import oracle as orc
    rng = np.random.default_rng(seed)
    tgt = rng.normal(size=(n_tgt, 3)) * 10
    qry = rng.normal(size=(n_qry, 3)) * 10
    idx, d2 = orc.KDTree(tgt).nearest_batch(qry)
    bi, bd = orc.nearest_batch_brute(tgt, qry)
    assert (idx == bi).all() and (d2 == bd).all()


@pytest.mark.parametrize("k", [1, 3, 20])
def test_k_nearest_matches_ckdtree(oracle, k):
    _, tgt, _ = synth.c1_room_corner(2000)
    tree = oracle.KDTree(tgt)
    ref = cKDTree(tgt)
    for q in range(0, 2000, 97):
        mine = tree.k_nearest(tgt[q], k)
        _, theirs = ref.query(tgt[q], k=k)
        assert (mine == np.atleast_1d(theirs)).all()
        assert mine[0] == q  # the query itself is in the tree (kdtree.hpp:144-180), distance 0
        assert (mine == oracle.k_nearest_brute(tgt, tgt[q], k)).all()


def test_k_nearest_fewer_points_than_k(oracle):
    tgt = np.array([[0.0, 0, 0], [1, 0, 0]])
    assert list(oracle.KDTree(tgt).k_nearest(np.array([0.1, 0, 0]), 20)) == [0, 1]
    n = oracle.estimate_normals(tgt, None, 20)
    assert (n == np.array([[0, 0, 1.0], [0, 0, 1.0]])).all()  # icp.hpp:34-37


def test_normals_match_eigh(oracle):
    _, tgt, _ = synth.c1_room_corner()
    nrm = oracle.estimate_normals(tgt, None, 20)
    _, nb = cKDTree(tgt).query(tgt, k=20)
    for i in range(0, tgt.shape[0], 41):
        P = tgt[nb[i]]
        c = P.mean(0)
        cov = (P - c).T @ (P - c) / 20
        w, v = np.linalg.eigh(cov)
        n = v[:, 0] if v[2, 0] >= 0 else -v[:, 0]
        gap = (w[1] - w[0]) / max(w[2], 1e-300)
        assert np.abs(n - nrm[i]).max() < 1e-13 / max(gap, 1e-6)
    np.testing.assert_allclose(np.linalg.norm(nrm, axis=1), 1.0, atol=1e-15)
    assert (nrm[:, 2] >= 0).all()  # icp.hpp:59-61


def test_normals_threads(oracle):
    _, tgt, _ = synth.c1_room_corner(1500)
    assert (oracle.estimate_normals(tgt, None, 20, nthreads=1)
            == oracle.estimate_normals(tgt, None, 20, nthreads=3)).all()


@settings(max_examples=40, deadline=None)
@given(st.integers(0, 2**31 - 1))
def test_smallest_eigenvector_property(seed):
    from oracle import oracle as orc
    rng = np.random.default_rng(seed)
    A = rng.normal(size=(3, 3))
    S = A @ A.T * rng.uniform(1e-6, 1e3)
    v = orc.smallest_eigenvector(S)
    w, V = np.linalg.eigh(S)
    assert abs(np.linalg.norm(v) - 1) < 1e-14
    gap = (w[1] - w[0]) / w[2]
    assert min(np.abs(v - V[:, 0]).max(), np.abs(v + V[:, 0]).max()) < 1e-13 / max(gap, 1e-9)


def test_smallest_eigenvector_diagonal(oracle):
    assert list(oracle.smallest_eigenvector(np.diag([3.0, 1.0, 2.0]))) == [0, 1, 0]
    assert list(oracle.smallest_eigenvector(np.zeros((3, 3)))) == [1, 0, 0]


def test_solve_matches_numpy(oracle):
    rng = np.random.default_rng(5)
    n = 500
    src = rng.normal(size=(n, 3)) * 5
    nrm = rng.normal(size=(n, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    T = synth.make_transform((0.01, 0.02, -0.015), (0.05, -0.02, 0.03))
    tgt = synth.apply_transform(T, src)
    got = oracle.solve_point_to_plane(src, tgt, nrm)
    J = np.hstack([np.cross(src, nrm), nrm])
    b = np.einsum("ij,ij->i", tgt - src, nrm)
    x = np.linalg.solve(J.T @ J, J.T @ b)
    want = np.eye(4)
    want[:3, :3] = Rotation.from_rotvec(x[:3]).as_matrix()
    want[:3, 3] = x[3:]
    np.testing.assert_allclose(got, want, atol=1e-12)
    sums = oracle.normal_equations(src, tgt, nrm)
    JtJ = J.T @ J
    np.testing.assert_allclose(sums[:21], JtJ[np.triu_indices(6)], rtol=1e-12)
    np.testing.assert_allclose(sums[21:27], J.T @ b, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(sums[27], b @ b, rtol=1e-12)


def test_solve_small_angle_identity_rotation(oracle):
    sums = np.zeros(28)
    sums[:21] = np.eye(6)[np.triu_indices(6)]
    sums[21:27] = [1e-12, 0, 0, 0.5, 0, 0]  # |r| < 1e-10 -> R = I (icp.hpp:130-131)
    T = oracle.solve_from_sums(sums)
    assert (T[:3, :3] == np.eye(3)).all() and T[0, 3] == 0.5


def test_ldlt_zero_pivot_rule(oracle):
    """KAT-5: a single plane z = 0 with normal (0,0,1) observes only rx, ry, tz; the
    pivoted LDLT restated from Eigen 3.4 maps |D_i| <= DBL_MIN to a zero component
    (pinned by construction, not by reference parity)."""
    rng = np.random.default_rng(9)
    src = np.c_[rng.uniform(-5, 5, (400, 2)), np.zeros(400)]
    nrm = np.tile([0.0, 0, 1.0], (400, 1))
    tgt = src + [0, 0, 0.01]
    T = oracle.solve_point_to_plane(src, tgt, nrm)
    assert np.isfinite(T).all()
    assert abs(T[2, 3] - 0.01) < 1e-12
    assert T[0, 3] == 0 and T[1, 3] == 0  # unobservable translation components -> 0


def test_voxel_downsample_matches_numpy_restatement(oracle):
    """file_utils.cpp:148-196: same voxel set and centroids as an independent numpy grouping."""
    rng = np.random.default_rng(3)
    pts = rng.uniform(-30, 30, (20000, 3))
    a = oracle.voxel_downsample(pts, 0.5)
    b = synth.voxel_centroids(pts, 0.5)
    assert a.shape == b.shape
    np.testing.assert_allclose(a, b, rtol=0, atol=1e-12)
    keys = np.floor(a / 0.5).astype(np.int64)
    assert len(np.unique(keys, axis=0)) == len(keys)          # one centroid per voxel
    assert (oracle.voxel_downsample(pts, 0.0) == pts).all()   # file_utils.cpp:152
    one = oracle.voxel_downsample(np.array([[0.1, 0.2, 0.3], [0.2, 0.1, 0.4]]), 0.5)
    np.testing.assert_allclose(one, [[0.15, 0.15, 0.35]], atol=1e-15)


def test_scan_context_against_numpy(oracle):
    """scan_context.hpp:44-82 and :90-142 against an independent numpy statement."""
    f0 = synth.lidar_frame(0, beams=32, azimuths=600)
    f1 = synth.lidar_frame(5, beams=32, azimuths=600)
    a, b = oracle.scan_context(f0), oracle.scan_context(f1)
    x, y, z = f0.T
    r = np.sqrt(x * x + y * y)
    ang = np.arctan2(y, x) + np.pi
    keep = ~((r > 80.0) | (r < 0.1))
    ri = np.clip((r / 4.0).astype(int), 0, 19)
    si = np.clip((ang / (2 * np.pi / 60)).astype(int), 0, 59)
    d = np.full((20, 60), -1e308)
    for k in np.nonzero(keep)[0]:
        d[ri[k], si[k]] = max(d[ri[k], si[k]], z[k])
    d[d < -1000] = 0
    assert (d == a).all()
    dist = min(1.0 - (a * np.roll(b, -s, axis=1)).sum() / (np.linalg.norm(a) * np.linalg.norm(b)) for s in range(60))
    assert abs(oracle.scan_context_distance(a, b) - dist) < 1e-12
    assert oracle.scan_context_distance(a, a) < 1e-15
    assert oracle.scan_context_distance(a, np.zeros((20, 60))) == 1.0


def test_oracle_reproduces_widening_golden(oracle):
    """tests/golden/widening.npz (scripts/make_golden.py): voxel filter and Scan Context of a raw
    synthetic scan -- guards the oracle's N1/N2 restatements against regressions."""
    import os
    import zlib
    from lidar_slam_from_scratch_amd import synth
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "widening.npz"))
    raw = synth.lidar_frame(0, voxel=0, beams=32, azimuths=900)
    assert np.uint32(zlib.crc32(np.ascontiguousarray(raw).tobytes())) == g["raw_crc"], "generator drifted"
    vox = oracle.voxel_downsample(raw, float(g["voxel_size"]))
    assert vox.shape[0] == int(g["voxel_rows"]) and (vox[:16] == g["voxel_head"]).all()
    assert np.uint32(zlib.crc32(np.ascontiguousarray(vox).tobytes())) == g["voxel_crc"]
    desc = oracle.scan_context(vox)
    assert (desc == g["sc_desc"]).all()
    assert oracle.scan_context_distance(desc, g["sc_desc_other"]) == float(g["sc_distance"])
