"""Host-side mirror of the reference's registration interface, backed by the HIP library.

Same names, argument meaning and result fields as the reference's C++ API
(slam_viz/include/slam_viz/core/):
    PointCloud            types.hpp:15-61
    Transformation        types.hpp:74-136
    ICPConfig             types.hpp:143-148
    ICPResult             types.hpp:155-164
    icp_point_to_plane    icp.hpp:157-258
    estimate_normals      icp.hpp:23-67
    solve_point_to_plane  icp.hpp:89-144
    KDTree                kdtree.hpp:18-186 (nearest, nearest_batch, k_nearest)
    NearestNeighborSearch kdtree.hpp:193-221
    ICP(...).align()      the facade BASELINE.json's north_star names

Every call goes through the C ABI (capi.Context) to the gfx950 kernels.  Error behaviour:
the reference defines none (empty clouds are UB there, kdtree.hpp:33-36); here an error
from the library raises capi.IcpError, and `icp_point_to_plane(..., on_error="unconverged")`
maps it to ICPResult(converged=False) so that the caller's gate (slam_node.cpp:139) yields
the identity, as SURVEY section 8b prescribes for the adapter.
"""
import numpy as np

from . import capi


class PointCloud:
    """N x 3 row-major fp64 cloud (types.hpp:15-61)."""

    def __init__(self, points=None):
        if points is None:
            points = np.zeros((0, 3))
        self._points = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)

    def points(self):
        return self._points

    def size(self):
        return self._points.shape[0]

    def empty(self):
        return self._points.shape[0] == 0

    def row(self, i):
        return self._points[i]

    def centroid(self):
        return self._points.mean(axis=0)

    def centered(self):
        return PointCloud(self._points - self._points.mean(axis=0))

    def copy(self):
        return PointCloud(self._points.copy())


class Transformation:
    """4x4 homogeneous rigid transform (types.hpp:74-136)."""

    def __init__(self, matrix=None):
        self._m = np.eye(4) if matrix is None else np.array(matrix, dtype=np.float64).reshape(4, 4)

    @staticmethod
    def from_rt(R, t):
        m = np.eye(4)
        m[:3, :3] = R
        m[:3, 3] = np.asarray(t, dtype=np.float64).reshape(3)
        return Transformation(m)

    @staticmethod
    def identity():
        return Transformation()

    def matrix(self):
        return self._m

    def R(self):
        return self._m[:3, :3].copy()

    def t(self):
        return self._m[:3, 3].copy()

    def apply(self, obj):
        if isinstance(obj, PointCloud):
            return PointCloud(obj.points() @ self._m[:3, :3].T + self._m[:3, 3])
        return self._m[:3, :3] @ np.asarray(obj, dtype=np.float64) + self._m[:3, 3]

    def compose(self, other):
        """this * other: this applied after other (types.hpp:118-120)."""
        return Transformation(self._m @ other._m)

    __mul__ = compose

    def inverse(self):
        Ri = self._m[:3, :3].T
        return Transformation.from_rt(Ri, -Ri @ self._m[:3, 3])


class ICPConfig:
    """types.hpp:143-148"""

    def __init__(self, max_iterations=50, tolerance=1e-6, min_error=1e-9, initial_transform=None):
        self.max_iterations = max_iterations
        self.tolerance = tolerance
        self.min_error = min_error
        self.initial_transform = initial_transform or Transformation.identity()


class ICPResult:
    """types.hpp:155-164"""

    def __init__(self):
        self.transformation = Transformation.identity()
        self.converged = False
        self.num_iterations = 0
        self.error_history = []
        self.final_error = 0.0

    def success(self):
        return self.converged and self.final_error < 0.1


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = capi.Context()
    return _default_ctx


def _pts(x):
    return x.points() if isinstance(x, PointCloud) else x


def icp_point_to_plane(source, target, config=None, ctx=None, on_error="raise"):
    """Point-to-plane ICP; returns T with T(source) ~ target (icp.hpp:157-258)."""
    config = config or ICPConfig()
    ctx = ctx or default_context()
    cfg = capi.Context.make_config(config.max_iterations, config.tolerance, config.min_error,
                                   config.initial_transform.matrix())
    out = ICPResult()
    try:
        res, hist = ctx.align(_pts(source), _pts(target), cfg)
    except capi.IcpError:
        if on_error == "unconverged":
            return out
        raise
    out.transformation = Transformation(np.array(res.transformation[:]).reshape(4, 4))
    out.converged = bool(res.converged)
    out.num_iterations = res.num_iterations
    out.error_history = list(hist)
    out.final_error = res.final_error
    return out


def estimate_normals(points, k=20, ctx=None):
    """icp.hpp:23-67 (the KD-tree argument of the reference is implicit: the search is exhaustive)."""
    return (ctx or default_context()).estimate_normals(_pts(points), k)


def solve_point_to_plane(source, target, normals, ctx=None):
    """icp.hpp:89-144"""
    return Transformation((ctx or default_context()).solve_point_to_plane(source, target, normals))


class KDTree:
    """kdtree.hpp:18-186 -- the name and the queries of the reference's tree; the search behind
    them is exhaustive on the GPU (same exact answers, no tree is built)."""

    def __init__(self, points, ctx=None):
        self._points = np.ascontiguousarray(_pts(points), dtype=np.float64)
        self._ctx = ctx or default_context()

    def size(self):
        return self._points.shape[0]

    def nearest(self, query):
        """kdtree.hpp:28-38 -> (index, squared distance)"""
        idx, d2 = self._ctx.nearest_batch(self._points, np.asarray(query, dtype=np.float64).reshape(1, 3))
        return int(idx[0]), float(d2[0])

    def nearest_batch(self, queries):
        """kdtree.hpp:43-59 -> (indices, distances_sq)"""
        return self._ctx.nearest_batch(self._points, _pts(queries))

    def k_nearest(self, query, k):
        """kdtree.hpp:65-78 -> indices of the k nearest points, closest first (at most size() of them)"""
        idx, _ = self._ctx.k_nearest(self._points, np.asarray(query, dtype=np.float64).reshape(1, 3), k, want_dist=False)
        return [int(j) for j in idx[0] if j >= 0]

    def k_nearest_batch(self, queries, k):
        """k_nearest for every row of `queries` -> (indices n x k, distances_sq n x k); -1 / inf where a list ends"""
        return self._ctx.k_nearest(self._points, _pts(queries), k)


class NearestNeighborSearch:
    """kdtree.hpp:193-221"""

    def __init__(self, target, ctx=None):
        self._target = np.ascontiguousarray(_pts(target), dtype=np.float64)
        self._ctx = ctx or default_context()

    def nearest_batch(self, queries):
        """kdtree.hpp:43-59 -> (indices, distances_sq)"""
        return self._ctx.nearest_batch(self._target, _pts(queries))

    def find_correspondences(self, source):
        """kdtree.hpp:198-214 -> (matched_target, distances)"""
        idx, d2 = self.nearest_batch(source)
        return self._target[idx], np.sqrt(d2)

    def tree(self):
        """kdtree.hpp:216"""
        return KDTree(self._target, self._ctx)


class ICP:
    """`ICP(config).align(source, target)` facade over icp_point_to_plane."""

    def __init__(self, config=None, ctx=None):
        self.config = config or ICPConfig()
        self.ctx = ctx

    def align(self, source, target):
        return icp_point_to_plane(source, target, self.config, self.ctx)
