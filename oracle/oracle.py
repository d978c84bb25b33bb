"""ctypes loader for the CPU oracle (oracle/icp_oracle.c).

TEST INFRASTRUCTURE ONLY: import this from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from the product package.  PARITY UNPINNED
(see icp_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    # ORACLE_SANITIZE=1: the AddressSanitizer + UBSan build (tests/test_oracle_residual_risk.py runs
    # the known-answer tests through it in a child process that preloads libasan)
    if os.environ.get("ORACLE_SANITIZE") == "1":
        subprocess.check_call(["make", "-s", "-C", _HERE, "libicp_oracle_asan.so"])
        return os.path.join(_HERE, "libicp_oracle_asan.so")
    so = os.path.join(_HERE, "libicp_oracle.so")
    src = os.path.join(_HERE, "icp_oracle.c")
    hdr = os.path.join(_HERE, "icp_oracle.h")
    stale = (not os.path.exists(so)) or any(
        os.path.exists(p) and os.path.getmtime(p) > os.path.getmtime(so) for p in (src, hdr))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "libicp_oracle.so"])
    return so


class Config(C.Structure):
    _fields_ = [("max_iterations", C.c_int), ("tolerance", C.c_double),
                ("min_error", C.c_double), ("initial_transform", C.c_double * 16)]


class Result(C.Structure):
    _fields_ = [("transformation", C.c_double * 16), ("converged", C.c_int),
                ("num_iterations", C.c_int), ("final_error", C.c_double),
                ("history_len", C.c_int), ("setup_seconds", C.c_double),
                ("loop_seconds", C.c_double), ("final_seconds", C.c_double),
                ("loop_iterations", C.c_int)]


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
        L.orc_kdtree_build.restype = C.c_void_p
        L.orc_kdtree_build.argtypes = [dp, C.c_int]
        L.orc_kdtree_free.argtypes = [C.c_void_p]
        L.orc_nearest_batch.argtypes = [C.c_void_p, dp, C.c_int, ip, dp, C.c_int]
        L.orc_k_nearest.restype = C.c_int
        L.orc_k_nearest.argtypes = [C.c_void_p, dp, C.c_int, ip]
        L.orc_nearest_batch_brute.argtypes = [dp, C.c_int, dp, C.c_int, ip, dp]
        L.orc_k_nearest_brute.restype = C.c_int
        L.orc_k_nearest_brute.argtypes = [dp, C.c_int, dp, C.c_int, ip]
        L.orc_estimate_normals.argtypes = [dp, C.c_int, C.c_void_p, C.c_int, dp, C.c_int]
        L.orc_estimate_normals_rows.argtypes = [dp, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, dp, C.c_int]
        L.orc_solve_point_to_plane.argtypes = [dp, dp, dp, C.c_int, dp]
        L.orc_normal_equations.argtypes = [dp, dp, dp, C.c_int, dp]
        L.orc_solve_from_sums.argtypes = [dp, dp]
        L.orc_smallest_eigenvector.argtypes = [dp, dp]
        L.orc_voxel_downsample.restype = C.c_int
        L.orc_voxel_downsample.argtypes = [dp, C.c_int, C.c_double, dp]
        L.orc_scan_context.argtypes = [dp, C.c_int, dp]
        L.orc_occupancy_cells.argtypes = [dp, C.c_int, dp, C.c_double, C.c_double, C.c_double, C.c_double, ip,
                                          C.POINTER(C.c_ubyte)]
        L.orc_scan_context_distance.restype = C.c_double
        L.orc_scan_context_distance.argtypes = [dp, dp]
        L.orc_icp_config_default.argtypes = [C.POINTER(Config)]
        L.orc_icp_point_to_plane.restype = C.c_int
        L.orc_icp_point_to_plane.argtypes = [dp, C.c_int, dp, C.c_int, C.POINTER(Config), C.c_int,
                                             C.c_int, C.c_int, C.POINTER(Result), dp, C.c_int]
        _LIB = L
    return _LIB


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(C.POINTER(C.c_double))


class KDTree:
    """kdtree.hpp:18-186"""

    def __init__(self, points):
        self.points, p = _d(points)
        self.n = self.points.shape[0]
        self._h = lib().orc_kdtree_build(p, self.n)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_kdtree_free(self._h)
            self._h = None

    def nearest_batch(self, queries, nthreads=1):
        q, qp = _d(queries)
        n = q.shape[0]
        idx = np.empty(n, dtype=np.int32)
        d2 = np.empty(n, dtype=np.float64)
        lib().orc_nearest_batch(self._h, qp, n, idx.ctypes.data_as(C.POINTER(C.c_int)),
                                d2.ctypes.data_as(C.POINTER(C.c_double)), nthreads)
        return idx, d2

    def k_nearest(self, query, k):
        q, qp = _d(query)
        out = np.empty(max(k, 1), dtype=np.int32)
        c = lib().orc_k_nearest(self._h, qp, k, out.ctypes.data_as(C.POINTER(C.c_int)))
        return out[:c].copy()


def nearest_batch_brute(targets, queries):
    t, tp = _d(targets)
    q, qp = _d(queries)
    n = q.shape[0]
    idx = np.empty(n, dtype=np.int32)
    d2 = np.empty(n, dtype=np.float64)
    lib().orc_nearest_batch_brute(tp, t.shape[0], qp, n, idx.ctypes.data_as(C.POINTER(C.c_int)),
                                  d2.ctypes.data_as(C.POINTER(C.c_double)))
    return idx, d2


def k_nearest_brute(targets, query, k):
    t, tp = _d(targets)
    q, qp = _d(query)
    out = np.empty(max(k, 1), dtype=np.int32)
    c = lib().orc_k_nearest_brute(tp, t.shape[0], qp, k, out.ctypes.data_as(C.POINTER(C.c_int)))
    return out[:c].copy()


def estimate_normals(points, tree=None, k=20, nthreads=1):
    """icp.hpp:23-67"""
    p, pp = _d(points)
    tree = tree or KDTree(p)
    out = np.empty_like(p)
    lib().orc_estimate_normals(pp, p.shape[0], tree._h, k, out.ctypes.data_as(C.POINTER(C.c_double)),
                               nthreads)
    return out


def estimate_normals_rows(points, row0, row1, tree=None, k=20, nthreads=1):
    """icp.hpp:23-67 for rows [row0, row1) only -> (row1 - row0) x 3"""
    p, pp = _d(points)
    tree = tree or KDTree(p)
    out = np.empty((max(0, row1 - row0), 3))
    lib().orc_estimate_normals_rows(pp, p.shape[0], tree._h, k, int(row0), int(row1),
                                    out.ctypes.data_as(C.POINTER(C.c_double)), nthreads)
    return out


def normal_equations(source, target, normals):
    s, sp = _d(source)
    t, tp = _d(target)
    n, np_ = _d(normals)
    out = np.empty(28)
    lib().orc_normal_equations(sp, tp, np_, s.shape[0], out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def solve_from_sums(sums):
    s, sp = _d(sums)
    T = np.empty(16)
    lib().orc_solve_from_sums(sp, T.ctypes.data_as(C.POINTER(C.c_double)))
    return T.reshape(4, 4)


def solve_point_to_plane(source, target, normals):
    """icp.hpp:89-144"""
    s, sp = _d(source)
    t, tp = _d(target)
    n, np_ = _d(normals)
    T = np.empty(16)
    lib().orc_solve_point_to_plane(sp, tp, np_, s.shape[0], T.ctypes.data_as(C.POINTER(C.c_double)))
    return T.reshape(4, 4)


def smallest_eigenvector(cov):
    c, cp = _d(np.asarray(cov).reshape(9))
    v = np.empty(3)
    lib().orc_smallest_eigenvector(cp, v.ctypes.data_as(C.POINTER(C.c_double)))
    return v


def voxel_downsample(points, voxel_size):
    """file_utils.cpp:148-196; voxels come out sorted by key (the reference's order is
    implementation-defined)."""
    p, pp = _d(points)
    out = np.empty_like(p)
    c = lib().orc_voxel_downsample(pp, p.shape[0], float(voxel_size), out.ctypes.data_as(C.POINTER(C.c_double)))
    return out[:c].copy()


GRID_DEFAULTS = dict(resolution=0.2, height_min=0.3, height_max=2.0, max_range=40.0)   # slam_node.hpp:35-40


def occupancy_cells(world, sensor, resolution=0.2, height_min=0.3, height_max=2.0, max_range=40.0):
    """update_occupancy_grid (slam_node.cpp:211-221): (cells int32 n x 2, keep bool n) per world point."""
    p, pp = _d(world)
    sx, sp = _d(np.asarray(sensor, dtype=np.float64).reshape(3))
    cells = np.zeros((p.shape[0], 2), dtype=np.int32)
    keep = np.zeros(p.shape[0], dtype=np.uint8)
    lib().orc_occupancy_cells(pp, p.shape[0], sp, float(resolution), float(height_min), float(height_max), float(max_range),
                              cells.ctypes.data_as(C.POINTER(C.c_int)), keep.ctypes.data_as(C.POINTER(C.c_ubyte)))
    return cells, keep.astype(bool)


def occupancy_update(cell_set, world, sensor, **grid):
    """occupied_cells_.insert(...) for every kept point: `cell_set` is a Python set of (x, y), updated in place."""
    cells, keep = occupancy_cells(world, sensor, **grid)
    cell_set.update(map(tuple, cells[keep].tolist()))
    return cell_set


def scan_context(cloud):
    """scan_context.hpp:44-82 -> 20 x 60 descriptor"""
    p, pp = _d(cloud)
    out = np.empty(1200)
    lib().orc_scan_context(pp, p.shape[0], out.ctypes.data_as(C.POINTER(C.c_double)))
    return out.reshape(20, 60)


def scan_context_distance(a, b):
    """scan_context.hpp:90-101"""
    a_ = np.ascontiguousarray(a, dtype=np.float64).reshape(1200)
    b_ = np.ascontiguousarray(b, dtype=np.float64).reshape(1200)
    return lib().orc_scan_context_distance(a_.ctypes.data_as(C.POINTER(C.c_double)),
                                           b_.ctypes.data_as(C.POINTER(C.c_double)))


class ICPResult:
    pass


def icp_point_to_plane(source, target, max_iterations=50, tolerance=1e-6, min_error=1e-9,
                       initial_transform=None, normal_k=20, faithful=True, nthreads=1):
    """icp.hpp:157-258.  Returns an object with the ICPResult fields (types.hpp:155-164)
    plus timing."""
    s, sp = _d(source)
    t, tp = _d(target)
    cfg = Config()
    lib().orc_icp_config_default(C.byref(cfg))
    cfg.max_iterations = int(max_iterations)
    cfg.tolerance = float(tolerance)
    cfg.min_error = float(min_error)
    if initial_transform is not None:
        it = np.ascontiguousarray(initial_transform, dtype=np.float64).reshape(16)
        for i in range(16):
            cfg.initial_transform[i] = it[i]
    cap = int(max_iterations) + 2
    hist = np.zeros(cap)
    res = Result()
    rc = lib().orc_icp_point_to_plane(sp, s.shape[0], tp, t.shape[0], C.byref(cfg), normal_k,
                                      1 if faithful else 0, nthreads, C.byref(res),
                                      hist.ctypes.data_as(C.POINTER(C.c_double)), cap)
    if rc != 0:
        raise ValueError("oracle icp: bad input")
    r = ICPResult()
    r.transformation = np.array(res.transformation[:]).reshape(4, 4)
    r.converged = bool(res.converged)
    r.num_iterations = res.num_iterations
    r.final_error = res.final_error
    r.error_history = hist[:res.history_len].copy()
    r.setup_seconds = res.setup_seconds
    r.loop_seconds = res.loop_seconds
    r.final_seconds = res.final_seconds
    r.loop_iterations = res.loop_iterations
    return r


# ---------------------------------------------------------------------------------------------
# Optional direct-oracle hook (oracle/ref_hook.cpp): the REFERENCE's own code behind a C wrapper,
# available only where Eigen3 and a reference checkout exist (not in the build image: `make ref`
# says so and builds nothing).  Used by tests/test_reference_hook.py to pin this oracle against
# the reference when it can be built; never by the product.
# ---------------------------------------------------------------------------------------------
_REF = None


def build_ref():
    """Path of oracle/_ref/libslam_ref.so, building it if possible; None when the reference
    cannot be built here (no Eigen3 / no checkout)."""
    so = os.path.join(_HERE, "_ref", "libslam_ref.so")
    if not os.path.exists(so):
        subprocess.call(["make", "-s", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return so if os.path.exists(so) else None


def ref_lib():
    global _REF
    if _REF is None:
        so = build_ref()
        if so is None:
            return None
        L = C.CDLL(so)
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
        L.ref_icp_point_to_plane.restype = C.c_int
        L.ref_icp_point_to_plane.argtypes = [dp, C.c_int, dp, C.c_int, C.c_int, C.c_double, C.c_double, dp,
                                             dp, ip, ip, dp, dp, C.c_int]
        L.ref_nearest_batch.argtypes = [dp, C.c_int, dp, C.c_int, ip, dp]
        L.ref_estimate_normals.argtypes = [dp, C.c_int, C.c_int, dp]
        L.ref_solve_point_to_plane.argtypes = [dp, dp, dp, C.c_int, dp]
        _REF = L
    return _REF


def ref_icp_point_to_plane(source, target, max_iterations=50, tolerance=1e-6, min_error=1e-9,
                           initial_transform=None):
    """slam::icp_point_to_plane itself -> (T 4x4, converged, num_iterations, final_error, history)."""
    L = ref_lib()
    s, sp = _d(source)
    t, tp = _d(target)
    T0 = np.ascontiguousarray(np.eye(4) if initial_transform is None else initial_transform, dtype=np.float64)
    T = np.zeros(16)
    conv, iters, ferr = C.c_int(0), C.c_int(0), C.c_double(0.0)
    hist = np.zeros(max_iterations + 2)
    n = L.ref_icp_point_to_plane(sp, s.shape[0], tp, t.shape[0], max_iterations, tolerance, min_error,
                                 T0.ctypes.data_as(C.POINTER(C.c_double)), T.ctypes.data_as(C.POINTER(C.c_double)),
                                 C.byref(conv), C.byref(iters), C.byref(ferr),
                                 hist.ctypes.data_as(C.POINTER(C.c_double)), hist.shape[0])
    return T.reshape(4, 4), bool(conv.value), iters.value, ferr.value, hist[:n].copy()


def ref_nearest_batch(targets, queries):
    L = ref_lib()
    t, tp = _d(targets)
    q, qp = _d(queries)
    idx = np.empty(q.shape[0], dtype=np.int32)
    d2 = np.empty(q.shape[0], dtype=np.float64)
    L.ref_nearest_batch(tp, t.shape[0], qp, q.shape[0], idx.ctypes.data_as(C.POINTER(C.c_int)),
                        d2.ctypes.data_as(C.POINTER(C.c_double)))
    return idx, d2


def ref_estimate_normals(points, k=20):
    L = ref_lib()
    p, pp = _d(points)
    out = np.empty_like(p)
    L.ref_estimate_normals(pp, p.shape[0], k, out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def ref_solve_point_to_plane(source, target, normals):
    L = ref_lib()
    s, sp = _d(source)
    t, tp = _d(target)
    n, np_ = _d(normals)
    T = np.zeros(16)
    L.ref_solve_point_to_plane(sp, tp, np_, s.shape[0], T.ctypes.data_as(C.POINTER(C.c_double)))
    return T.reshape(4, 4)
