"""The three places where the oracle restates arithmetic that lives inside Eigen (DESIGN.md
section 2: cyclic Jacobi for SelfAdjointEigenSolver, an index-order pivoted LDLT for
Eigen::LDLT, own k-NN selection) exercised where they are most likely to part from the
reference: ill-conditioned inputs.  The reference holds no fixtures and Eigen is not in this
image, so these checks are against NumPy / SciPy (independent, third-party) -- the oracle stays
"parity unpinned"; what they pin is that the restatement is a correct solver of the same
problems on the data far-range LiDAR rings produce.  Plus the sanitizer run of the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

from lidar_slam_from_scratch_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def ring_cloud(rings=8, per_ring=1800, seed=0):
    """Ground returns of the `rings` lowest-but-far beams of a 64-beam spinning LiDAR (sensor
    1.73 m above a flat ground): each beam draws one circle, sampled every 0.2 degrees with
    centimetre range noise.  A point's 20 nearest neighbours are its own arc: a nearly collinear
    set whose covariance has two tiny, nearly equal eigenvalues."""
    rng = np.random.default_rng(seed)
    h = 1.73
    elev = np.deg2rad(np.linspace(-2.6, -1.4, rings))          # far range: 38 m .. 71 m
    az = np.deg2rad(np.arange(per_ring) * (360.0 / per_ring))
    pts = []
    for e in elev:
        r = h / np.tan(-e) + rng.normal(0, 0.01, per_ring)     # range noise ~ 1 cm
        z = -h + rng.normal(0, 0.003, per_ring)
        pts.append(np.stack([r * np.cos(az), r * np.sin(az), z], axis=1))
    return np.ascontiguousarray(np.vstack(pts))


def neighbourhood_cov(points, idx):
    nb = points[idx]
    c = nb.mean(axis=0)
    d = nb - c
    return d.T @ d / len(idx)


def test_ring_normals_are_smallest_eigen_directions(oracle):
    """Jacobi vs numpy.linalg.eigh on ring neighbourhoods.  Where the two smallest eigenvalues
    are separated, the directions agree to (machine epsilon x norm / gap); where they are not,
    `the' smallest eigenvector is not defined to better than that either, and the check is the
    one that always holds: the returned unit vector's Rayleigh quotient is the smallest
    eigenvalue to rounding.  This is the property the ICP consumes (the residual direction)."""
    pts = ring_cloud(rings=4, per_ring=1800, seed=3)
    normals = oracle.estimate_normals(pts, None, 20, nthreads=4)
    tree = oracle.KDTree(pts)
    checked_gap = 0
    for i in range(0, pts.shape[0], 37):
        C = neighbourhood_cov(pts, tree.k_nearest(pts[i], 20))
        w, V = np.linalg.eigh(C)
        v = normals[i]
        assert abs(np.linalg.norm(v) - 1.0) < 1e-14 and v[2] >= 0.0               # icp.hpp:59-63
        scale = np.abs(C).max()
        assert v @ C @ v <= w[0] + 64 * np.finfo(float).eps * scale, (i, w)
        gap = w[1] - w[0]
        if gap > 1e-9 * scale:
            ang = np.linalg.norm(np.cross(v, V[:, 0]))  # sine of the angle: arccos near 1 resolves 1e-8 only
            assert ang <= 256 * np.finfo(float).eps * scale / gap + 1e-12, (i, ang, gap / scale)
            checked_gap += 1
    assert checked_gap > 50
    # and the eigen-direction itself, without the neighbour search, on matrices with a tiny gap
    rng = np.random.default_rng(11)
    for rel_gap in (1e-3, 1e-6, 1e-9, 1e-12):
        Q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        C = Q @ np.diag([1e-4, 1e-4 * (1 + rel_gap), 25.0]) @ Q.T
        C = 0.5 * (C + C.T)
        v = oracle.smallest_eigenvector(C)
        w, V = np.linalg.eigh(C)
        assert abs(v @ C @ v - w[0]) <= 64 * np.finfo(float).eps * 25.0
        assert abs(np.linalg.norm(v) - 1.0) < 1e-14


def two_plane_system(noise, seed):
    """Correspondences on two planes only (a street with one facade): J^T J has one direction --
    the translation along the planes' common line -- that only the noise constrains."""
    rng = np.random.default_rng(seed)
    n = 4000
    a = np.stack([rng.uniform(-20, 20, n // 2), rng.uniform(-20, 20, n // 2), np.zeros(n // 2)], axis=1)   # ground z = 0
    b = np.stack([rng.uniform(-20, 20, n // 2), np.full(n // 2, 8.0), rng.uniform(0, 6, n // 2)], axis=1)  # wall y = 8
    tgt = np.vstack([a, b])
    nrm = np.vstack([np.tile([0.0, 0.0, 1.0], (n // 2, 1)), np.tile([0.0, 1.0, 0.0], (n // 2, 1))])
    nrm = nrm + rng.normal(0, noise, nrm.shape)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    T = synth.make_transform((0.004, -0.003, 0.005), (0.03, -0.02, 0.01))
    src = (tgt - T[:3, 3]) @ T[:3, :3]
    return src, tgt, nrm


def unpack(sums):
    A = np.zeros((6, 6))
    o = 0
    for r in range(6):
        for c in range(r, 6):
            A[r, c] = A[c, r] = sums[o]
            o += 1
    return A, sums[21:27].copy()


@pytest.mark.parametrize("noise", [1e-2, 1e-4, 1e-6])
def test_near_singular_normal_equations(oracle, noise):
    """Pivoted LDLT (icp.hpp:120) on two-plane geometry: condition number 1e4 .. 1e12.  The
    update must solve the system (small residual) and agree with LAPACK's solution to within
    what the conditioning allows; SciPy's Bunch-Kaufman LDL^T of the same matrix must agree on
    the inertia (all pivots positive: no direction was dropped by the zero-pivot rule)."""
    from scipy.linalg import ldl
    from scipy.spatial.transform import Rotation
    src, tgt, nrm = two_plane_system(noise, seed=5)
    sums = oracle.normal_equations(src, tgt, nrm)
    A, b = unpack(sums)
    cond = np.linalg.cond(A)
    assert 1e3 < cond < 1e14
    T = oracle.solve_from_sums(np.concatenate([sums, [float(src.shape[0])]]) if sums.shape[0] == 28 else sums)
    x = np.concatenate([Rotation.from_matrix(T[:3, :3]).as_rotvec(), T[:3, 3]])
    ref = np.linalg.solve(A, b)
    eps = np.finfo(float).eps
    assert np.linalg.norm(A @ x - b) <= 1e3 * eps * np.linalg.norm(A, 2) * np.linalg.norm(x) + 1e-18
    assert np.linalg.norm(x - ref) <= 1e3 * cond * eps * np.linalg.norm(ref) + 1e-15
    _, d, _ = ldl(A)
    assert (np.diag(d) > 0).all() and np.count_nonzero(d - np.diag(np.diag(d))) == 0


def test_plane_plus_line_leaves_the_free_direction_alone(oracle):
    """A plane and one line of points on it constrain 4 of the 6 parameters only when the line's
    normals lie in the plane's normal: exactly singular.  Eigen's rule (LDLT::_solve_impl: a pivot
    with |D_i| <= DBL_MIN contributes 0) restated: the solution has no component along the null
    space beyond rounding, and what is constrained matches the least-squares solution."""
    rng = np.random.default_rng(2)
    n = 3000
    tgt = np.stack([rng.uniform(-10, 10, n), rng.uniform(-10, 10, n), np.zeros(n)], axis=1)
    nrm = np.tile([0.0, 0.0, 1.0], (n, 1))
    T = synth.make_transform((0.002, -0.001, 0.0), (0.0, 0.0, 0.05))
    src = (tgt - T[:3, 3]) @ T[:3, :3]
    sums = oracle.normal_equations(src, tgt, nrm)
    A, b = unpack(sums)
    assert np.linalg.matrix_rank(A, tol=1e-9 * np.abs(A).max()) == 3
    Tm = oracle.solve_from_sums(sums)
    from scipy.spatial.transform import Rotation
    x = np.concatenate([Rotation.from_matrix(Tm[:3, :3]).as_rotvec(), Tm[:3, 3]])
    ls = np.linalg.lstsq(A, b, rcond=1e-12)[0]
    # observable part: rx, ry, tz (rotation about the in-plane axes, translation along the normal)
    np.testing.assert_allclose(x[[0, 1, 5]], ls[[0, 1, 5]], rtol=1e-9, atol=1e-13)
    assert np.abs(x[[2, 3, 4]]).max() <= 1e-9     # rz, tx, ty: unobservable, left at zero
    assert np.isfinite(x).all()


def test_oracle_under_address_and_ub_sanitizers():
    """SURVEY section 5 (sanitizer row): the known-answer paths of the oracle under
    -fsanitize=address,undefined, in a child process (the interpreter itself is not instrumented,
    so libasan is preloaded; leaks of the interpreter are not our concern)."""
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], stdout=subprocess.PIPE, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan in this toolchain")
    code = r"""
import sys; sys.path.insert(0, %r)
import numpy as np
from oracle import oracle as orc
from lidar_slam_from_scratch_amd import synth
src, tgt, T = synth.c1_room_corner(1500)
tree = orc.KDTree(tgt)
i, d = tree.nearest_batch(src, nthreads=2)
bi, bd = orc.nearest_batch_brute(tgt, src)
assert (i == bi).all() and (d == bd).all()
assert (tree.k_nearest(tgt[7], 20) == orc.k_nearest_brute(tgt, tgt[7], 20)).all()
assert len(orc.KDTree(tgt[:5]).k_nearest(tgt[0], 20)) == 5
n = orc.estimate_normals(tgt, tree, 20, nthreads=2)
assert np.allclose(np.linalg.norm(n, axis=1), 1.0)
r = orc.icp_point_to_plane(src, tgt)
assert r.converged and synth.pose_delta(r.transformation, T)[0] < 1e-2
r = orc.icp_point_to_plane(src, tgt, max_iterations=3, tolerance=0.0, min_error=0.0, faithful=False, nthreads=2)
assert r.num_iterations == 3
v = orc.voxel_downsample(np.vstack([src, -src]), 0.5)
assert 0 < v.shape[0] <= 2 * src.shape[0]
sc = orc.scan_context(src)
assert orc.scan_context_distance(sc, sc) < 1e-12
cells, keep = orc.occupancy_cells(np.vstack([src, [[np.nan, 1, 1], [1e300, 0, 1], [4e8, 4e8, 1], [3, np.inf, 1]]]), [0.3, -0.2, 0.0])
assert keep[:src.shape[0]].any() and not keep[-4:].any()      # (casts of non-finite / huge quotients must not be reached)
plane = np.stack([src[:, 0], src[:, 1], 0 * src[:, 2]], axis=1)
orc.solve_point_to_plane(plane, plane, np.tile([0., 0., 1.], (plane.shape[0], 1)))   # singular 6x6: zero-pivot rule
print("sanitized oracle ok")
""" % ROOT
    env = dict(os.environ, ORACLE_SANITIZE="1", LD_PRELOAD=asan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=66", UBSAN_OPTIONS="halt_on_error=1:exitcode=67")
    out = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                         timeout=600)
    assert out.returncode == 0 and "sanitized oracle ok" in out.stdout, out.stdout[-4000:]
    assert "AddressSanitizer" not in out.stdout and "runtime error" not in out.stdout, out.stdout[-4000:]
