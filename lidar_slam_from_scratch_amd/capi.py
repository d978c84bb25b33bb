"""ctypes binding of include/icp_mi355x.h (the C ABI of libicp_mi355x.so).

The library is the product; there is deliberately no Python or CPU fallback here:
if the shared object is missing or no gfx950 device is visible, calls raise.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

OK = 0
ERR_NULL, ERR_EMPTY_SOURCE, ERR_EMPTY_TARGET, ERR_CAPACITY = -1, -2, -3, -4
ERR_HIP, ERR_RCCL, ERR_ARG, ERR_NO_DEVICE = -5, -6, -7, -8
SEARCH_AUTO, SEARCH_EXACT_F64, SEARCH_MFMA_BF16, SEARCH_MFMA_PRUNED = 0, 1, 2, 3
UNIQUE_ID_BYTES = 128

EXPORTS = [
    "icpmi_version", "icpmi_options_default", "icpmi_config_default", "icpmi_create",
    "icpmi_destroy", "icpmi_last_error", "icpmi_align", "icpmi_align_device", "icpmi_align_batch",
    "icpmi_nearest_batch", "icpmi_k_nearest", "icpmi_estimate_normals", "icpmi_solve_point_to_plane",
    "icpmi_transform_points", "icpmi_comm_unique_id", "icpmi_comm_init", "icpmi_comm_finalize",
    "icpmi_comm_init_callbacks", "icpmi_comm_info", "icpmi_voxel_downsample", "icpmi_voxel_downsample_device",
    "icpmi_scan_context", "icpmi_scan_context_distances", "icpmi_load_cloud", "icpmi_load_cloud_device",
    "icpmi_upload_points_f32", "icpmi_discover_frames", "icpmi_estimate_normals_rows",
    "icpmi_stream_push", "icpmi_stream_push_host", "icpmi_stream_push_file", "icpmi_stream_prefetch_file", "icpmi_stream_reset",
    "icpmi_grid_config_default", "icpmi_occupancy_update", "icpmi_occupancy_update_device", "icpmi_occupancy_cells",
    "icpmi_occupancy_clear", "icpmi_stream_map_update", "icpmi_stream_current_scan",
    "icpmi_reset_profile", "icpmi_get_profile",
]


class Options(C.Structure):
    _fields_ = [("device", C.c_int32), ("normal_k", C.c_int32), ("search", C.c_int32),
                ("profile", C.c_int32)]


class Config(C.Structure):
    _fields_ = [("max_iterations", C.c_int32), ("reserved", C.c_int32),
                ("tolerance", C.c_double), ("min_error", C.c_double),
                ("initial_transform", C.c_double * 16)]


class Result(C.Structure):
    _fields_ = [("transformation", C.c_double * 16), ("converged", C.c_int32),
                ("num_iterations", C.c_int32), ("final_error", C.c_double),
                ("history_len", C.c_int32), ("loop_iterations", C.c_int32)]


class Profile(C.Structure):
    _fields_ = [("nn_ms", C.c_double), ("nn_launches", C.c_int64),
                ("coarse_ms", C.c_double), ("coarse_launches", C.c_int64),
                ("reduce_ms", C.c_double), ("reduce_launches", C.c_int64),
                ("transform_ms", C.c_double), ("transform_launches", C.c_int64),
                ("normals_ms", C.c_double), ("normals_launches", C.c_int64),
                ("total_ms", C.c_double), ("calls", C.c_int64), ("loop_ms", C.c_double),
                ("setup_ms", C.c_double),
                ("nn_pairs", C.c_double), ("nn_recheck_queries", C.c_int64),
                ("nn_fallback_queries", C.c_int64), ("knn_fallback_rows", C.c_int64),
                ("nn_coarse_blocks", C.c_int64), ("nn_pruned_blocks", C.c_int64), ("small_launches", C.c_int64),
                ("bounded_launches", C.c_int64),
                ("nn_group_pairs", C.c_int64),
                ("nn_group_pairs_run", C.c_int64),
                ("exchange_ms", C.c_double), ("exchange_launches", C.c_int64), ("coarse_minima_bytes", C.c_int64)]


MAX_RANKS_INFO = 64
MAX_BATCH = 8   # ICPMI_MAX_BATCH


class CommInfo(C.Structure):
    _fields_ = [("kind", C.c_int32), ("n_ranks", C.c_int32), ("rank", C.c_int32), ("reserved", C.c_int32),
                ("device", C.c_int32 * MAX_RANKS_INFO), ("pci_bus_id", (C.c_char * 16) * MAX_RANKS_INFO)]


class StreamInfo(C.Structure):
    _fields_ = [("status", C.c_int32), ("reserved", C.c_int32), ("n_filtered", C.c_int64), ("n_target", C.c_int64)]


STREAM_REGISTERED, STREAM_FIRST_FRAME, STREAM_TOO_FEW_POINTS = 0, 1, 2


class GridConfig(C.Structure):
    """OccupancyGridConfig (slam_node.hpp:35-40)"""
    _fields_ = [("resolution", C.c_double), ("height_min", C.c_double), ("height_max", C.c_double), ("max_range", C.c_double)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int32)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int32)


class IcpError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("icp_mi355x error %d: %s" % (code, message))
        self.code = code


_LIB = None


def hip_runtimes_mapped(maps_text=None):
    """Distinct libamdhip64 images mapped into this process (real paths), from /proc/self/maps."""
    if maps_text is None:
        try:
            with open("/proc/self/maps") as f:
                maps_text = f.read()
        except OSError:
            return []
    seen = []
    for line in maps_text.splitlines():
        parts = line.split(None, 5)
        if len(parts) == 6 and os.path.basename(parts[5]).startswith("libamdhip64"):
            path = parts[5].strip()
            if path not in seen:
                seen.append(path)
    return seen


def _one_hip_runtime(before_dlopen):
    """One HIP runtime per process.  libicp_mi355x.so needs `libamdhip64.so.7`; the torch wheel
    bundles its own copy (SONAME libamdhip64.so.7, but torch's libraries ask for it as
    `libamdhip64.so`, which the dynamic loader does not recognise as the image /opt/rocm already
    provided).  Library first, torch second therefore maps TWO runtimes, and the first stream or
    event handed from one to the other aborts the process (`std::bad_variant_access`).  Torch
    first, library second resolves both to torch's copy.  So: import torch before the dlopen if
    it is there to be imported, and refuse to go on if two runtimes are mapped anyway."""
    import sys
    if before_dlopen:
        if "torch" not in sys.modules:
            import importlib.util
            if importlib.util.find_spec("torch") is not None:
                import torch  # noqa: F401
        return
    mapped = hip_runtimes_mapped()
    if len(mapped) > 1:
        raise IcpError(ERR_HIP, "two HIP runtimes are mapped into this process (%s): libicp_mi355x.so must "
                                "share torch's; import torch before anything loads /opt/rocm's libamdhip64"
                       % ", ".join(mapped))


def load_library(path=None):
    """dlopen the in-tree libicp_mi355x.so (must have been built: see build.build_library)."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    path = path or _build.LIB_PATH
    if not os.path.exists(path):
        raise FileNotFoundError(
            "%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback)" % path)
    _one_hip_runtime(before_dlopen=True)
    L = C.CDLL(path)
    _one_hip_runtime(before_dlopen=False)
    dp, vp = C.POINTER(C.c_double), C.c_void_p
    L.icpmi_version.restype = C.c_char_p
    L.icpmi_options_default.argtypes = [C.POINTER(Options)]
    L.icpmi_config_default.argtypes = [C.POINTER(Config)]
    L.icpmi_create.argtypes = [C.POINTER(Options), C.POINTER(vp)]
    L.icpmi_destroy.argtypes = [vp]
    L.icpmi_destroy.restype = None
    L.icpmi_last_error.argtypes = [vp]
    L.icpmi_last_error.restype = C.c_char_p
    L.icpmi_align.argtypes = [vp, dp, C.c_int64, dp, C.c_int64, C.POINTER(Config),
                              C.POINTER(Result), dp, C.c_int32]
    L.icpmi_align_device.argtypes = [vp, vp, C.c_int64, vp, C.c_int64, C.POINTER(Config),
                                     C.POINTER(Result), dp, C.c_int32]
    L.icpmi_align_batch.argtypes = [vp, C.c_int32, C.POINTER(dp), C.POINTER(C.c_int64), C.POINTER(dp), C.POINTER(C.c_int64),
                                    C.POINTER(Config), C.POINTER(Result), dp, C.c_int32, C.POINTER(C.c_int32)]
    L.icpmi_nearest_batch.argtypes = [vp, dp, C.c_int64, dp, C.c_int64, C.POINTER(C.c_int32), dp]
    L.icpmi_k_nearest.argtypes = [vp, dp, C.c_int64, dp, C.c_int64, C.c_int32, C.POINTER(C.c_int32), dp]
    L.icpmi_estimate_normals.argtypes = [vp, dp, C.c_int64, C.c_int32, dp]
    L.icpmi_solve_point_to_plane.argtypes = [vp, dp, dp, dp, C.c_int64, dp]
    L.icpmi_transform_points.argtypes = [vp, dp, dp, C.c_int64, dp]
    i64p = C.POINTER(C.c_int64)
    L.icpmi_voxel_downsample.argtypes = [vp, dp, C.c_int64, C.c_double, dp, C.c_int64, i64p]
    L.icpmi_voxel_downsample_device.argtypes = [vp, vp, C.c_int64, C.c_double, vp, C.c_int64, i64p]
    L.icpmi_load_cloud.argtypes = [C.c_char_p, dp, C.c_int64, i64p]
    L.icpmi_load_cloud_device.argtypes = [vp, C.c_char_p, vp, C.c_int64, i64p]
    L.icpmi_upload_points_f32.argtypes = [vp, C.POINTER(C.c_float), C.c_int64, C.c_int32, vp]
    L.icpmi_discover_frames.argtypes = [C.c_char_p, i64p, C.c_int64, C.c_char_p, C.c_int64, i64p, i64p]
    L.icpmi_estimate_normals_rows.argtypes = [vp, dp, C.c_int64, C.c_int32, C.c_int64, C.c_int64, dp]
    L.icpmi_stream_push.argtypes = [vp, vp, C.c_int64, C.c_double, C.c_int64, C.POINTER(Config), C.POINTER(Result), dp,
                                    C.c_int32, C.POINTER(StreamInfo)]
    L.icpmi_stream_push_host.argtypes = [vp, dp, C.c_int64, C.c_double, C.c_int64, C.POINTER(Config), C.POINTER(Result), dp,
                                         C.c_int32, C.POINTER(StreamInfo)]
    L.icpmi_stream_push_file.argtypes = [vp, C.c_char_p, C.c_double, C.c_int64, C.POINTER(Config), C.POINTER(Result), dp,
                                         C.c_int32, C.POINTER(StreamInfo)]
    L.icpmi_stream_reset.argtypes = [vp]
    L.icpmi_stream_prefetch_file.argtypes = [vp, C.c_char_p]
    L.icpmi_grid_config_default.argtypes = [C.POINTER(GridConfig)]
    L.icpmi_grid_config_default.restype = None
    L.icpmi_occupancy_update.argtypes = [vp, dp, C.c_int64, dp, C.POINTER(GridConfig), i64p]
    L.icpmi_occupancy_update_device.argtypes = [vp, vp, C.c_int64, dp, C.POINTER(GridConfig), i64p]
    L.icpmi_occupancy_cells.argtypes = [vp, C.POINTER(C.c_int32), C.c_int64, i64p]
    L.icpmi_occupancy_clear.argtypes = [vp]
    L.icpmi_stream_map_update.argtypes = [vp, dp, C.POINTER(GridConfig), dp, C.c_int64, i64p, i64p]
    L.icpmi_stream_current_scan.argtypes = [vp, dp, C.c_int64, i64p]
    L.icpmi_scan_context.argtypes = [vp, dp, C.c_int64, dp]
    L.icpmi_scan_context_distances.argtypes = [vp, dp, dp, C.c_int64, dp]
    L.icpmi_comm_unique_id.argtypes = [vp, vp]
    L.icpmi_comm_init.argtypes = [vp, C.c_int32, C.c_int32, vp]
    L.icpmi_comm_finalize.argtypes = [vp]
    L.icpmi_comm_init_callbacks.argtypes = [vp, C.c_int32, C.c_int32, ALLREDUCE_FN, ALLGATHER_FN, vp]
    L.icpmi_comm_info.argtypes = [vp, C.POINTER(CommInfo)]
    L.icpmi_reset_profile.argtypes = [vp]
    L.icpmi_get_profile.argtypes = [vp, C.POINTER(Profile)]
    for name in EXPORTS:
        getattr(L, name)  # raises AttributeError if a declared symbol is not exported
    _LIB = L
    return L


def load_cloud(path):
    """load_ply / load_bin (file_utils.cpp:20-141) through the library: N x 3 fp64."""
    L = load_library()
    n = C.c_int64(0)
    rc = L.icpmi_load_cloud(os.fsencode(path), None, 0, C.byref(n))
    if rc != OK:
        raise IcpError(rc, L.icpmi_last_error(None).decode())
    out = np.empty((n.value, 3))
    rc = L.icpmi_load_cloud(os.fsencode(path), out.ctypes.data_as(C.POINTER(C.c_double)), n.value, C.byref(n))
    if rc != OK:
        raise IcpError(rc, L.icpmi_last_error(None).decode())
    return out


def discover_frames(data_dir):
    """discover_frames (file_utils.cpp:217-247) through the library -> [(number, path), ...] sorted by number."""
    L = load_library()
    n, nbytes = C.c_int64(0), C.c_int64(0)
    rc = L.icpmi_discover_frames(os.fsencode(data_dir), None, 0, None, 0, C.byref(n), C.byref(nbytes))
    if rc != OK:
        raise IcpError(rc, L.icpmi_last_error(None).decode())
    if n.value == 0:
        return []
    stamps = (C.c_int64 * n.value)()
    buf = C.create_string_buffer(nbytes.value)
    rc = L.icpmi_discover_frames(os.fsencode(data_dir), stamps, n.value, buf, nbytes.value, C.byref(n), C.byref(nbytes))
    if rc != OK:
        raise IcpError(rc, L.icpmi_last_error(None).decode())
    paths = buf.raw[:nbytes.value].split(b"\0")[:n.value]
    return [(int(stamps[i]), os.fsdecode(paths[i])) for i in range(n.value)]


def _f64(a, cols=3):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if cols and (a.ndim != 2 or a.shape[1] != cols):
        raise ValueError("expected an N x %d array" % cols)
    return a


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Context:
    """One icpmi_ctx: a device, a stream and its workspace."""

    def __init__(self, device=0, normal_k=20, search=SEARCH_AUTO, profile=False):
        self._lib = load_library()
        opt = Options()
        self._lib.icpmi_options_default(C.byref(opt))
        opt.device, opt.normal_k, opt.search, opt.profile = device, normal_k, search, int(profile)
        h = C.c_void_p()
        rc = self._lib.icpmi_create(C.byref(opt), C.byref(h))
        if rc != OK:
            raise IcpError(rc, self._lib.icpmi_last_error(None).decode())
        self._h = h
        self.n_ranks, self.rank = 1, 0

    def close(self):
        if getattr(self, "_h", None):
            self._lib.icpmi_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc):
        if rc != OK:
            raise IcpError(rc, self._lib.icpmi_last_error(self._h).decode())

    @staticmethod
    def make_config(max_iterations=50, tolerance=1e-6, min_error=1e-9, initial_transform=None):
        cfg = Config()
        cfg.max_iterations, cfg.tolerance, cfg.min_error = int(max_iterations), tolerance, min_error
        it = np.eye(4) if initial_transform is None else np.asarray(initial_transform, dtype=np.float64)
        for i, v in enumerate(it.reshape(16)):
            cfg.initial_transform[i] = v
        return cfg

    def align(self, source, target, cfg):
        src, tgt = _f64(source), _f64(target)
        cap = cfg.max_iterations + 1
        hist = np.zeros(max(cap, 1))
        res = Result()
        self._check(self._lib.icpmi_align(self._h, _dp(src), src.shape[0], _dp(tgt), tgt.shape[0],
                                          C.byref(cfg), C.byref(res), _dp(hist), cap))
        return res, hist[:res.history_len].copy()

    def align_batch(self, sources, targets, cfgs):
        """Several independent registrations at once (icpmi_align_batch: the verifications of one
        LoopClosureDetector::detect, loop_closure.hpp:94-123) -> [(Result, error history), ...], each
        bit-identical to align() of the same pair.  cfgs: one Config for all, or one per pair."""
        srcs, tgts = [_f64(a) for a in sources], [_f64(a) for a in targets]
        k = len(srcs)
        if k != len(tgts) or k < 1:
            raise ValueError("as many targets as sources, at least one")
        cfg_list = list(cfgs) if isinstance(cfgs, (list, tuple)) else [cfgs] * k
        cfg_arr = (Config * k)(*cfg_list)
        stride = max(c.max_iterations for c in cfg_list) + 1
        hist = np.zeros((k, stride))
        res = (Result * k)()
        status = (C.c_int32 * k)()
        dpp = C.POINTER(C.c_double)
        sp = (dpp * k)(*[_dp(a) for a in srcs])
        tp = (dpp * k)(*[_dp(a) for a in tgts])
        ns = (C.c_int64 * k)(*[a.shape[0] for a in srcs])
        nt = (C.c_int64 * k)(*[a.shape[0] for a in tgts])
        self._check(self._lib.icpmi_align_batch(self._h, k, sp, ns, tp, nt, cfg_arr, res, _dp(hist), stride, status))
        out = []
        for i in range(k):
            r = Result()
            C.memmove(C.byref(r), C.byref(res[i]), C.sizeof(Result))
            out.append((r, hist[i, :r.history_len].copy()))
        return out

    def align_device(self, src_ptr, n_src, tgt_ptr, n_tgt, cfg):
        """src_ptr/tgt_ptr: device addresses of row-major N x 3 fp64 (e.g. tensor.data_ptr())."""
        cap = cfg.max_iterations + 1
        hist = np.zeros(max(cap, 1))
        res = Result()
        self._check(self._lib.icpmi_align_device(self._h, C.c_void_p(src_ptr), n_src,
                                                 C.c_void_p(tgt_ptr), n_tgt, C.byref(cfg),
                                                 C.byref(res), _dp(hist), cap))
        return res, hist[:res.history_len].copy()

    def nearest_batch(self, targets, queries, want_dist=True):
        tgt, qry = _f64(targets), _f64(queries)
        idx = np.empty(qry.shape[0], dtype=np.int32)
        d2 = np.empty(qry.shape[0]) if want_dist else None
        self._check(self._lib.icpmi_nearest_batch(
            self._h, _dp(tgt), tgt.shape[0], _dp(qry), qry.shape[0],
            idx.ctypes.data_as(C.POINTER(C.c_int32)), _dp(d2) if want_dist else None))
        return idx, d2

    def k_nearest(self, targets, queries, k, want_dist=True):
        """kdtree.hpp:65-78 for a batch of queries -> (indices n x k closest first, squared distances)"""
        tgt, qry = _f64(targets), _f64(queries)
        idx = np.empty((qry.shape[0], k), dtype=np.int32)
        d2 = np.empty((qry.shape[0], k)) if want_dist else None
        self._check(self._lib.icpmi_k_nearest(
            self._h, _dp(tgt), tgt.shape[0], _dp(qry), qry.shape[0], k,
            idx.ctypes.data_as(C.POINTER(C.c_int32)), _dp(d2) if want_dist else None))
        return idx, d2

    def estimate_normals(self, points, k=20):
        pts = _f64(points)
        out = np.empty_like(pts)
        self._check(self._lib.icpmi_estimate_normals(self._h, _dp(pts), pts.shape[0], k, _dp(out)))
        return out

    def solve_point_to_plane(self, source, target, normals):
        s, t, n = _f64(source), _f64(target), _f64(normals)
        if not (s.shape == t.shape == n.shape):
            raise ValueError("source, target and normals must have the same shape")
        T = np.empty(16)
        self._check(self._lib.icpmi_solve_point_to_plane(self._h, _dp(s), _dp(t), _dp(n), s.shape[0], _dp(T)))
        return T.reshape(4, 4)

    def transform_points(self, T, points):
        pts = _f64(points)
        Tm = np.ascontiguousarray(T, dtype=np.float64).reshape(16)
        out = np.empty_like(pts)
        self._check(self._lib.icpmi_transform_points(self._h, _dp(Tm), _dp(pts), pts.shape[0], _dp(out)))
        return out

    def voxel_downsample(self, points, voxel_size):
        """file_utils.cpp:148-196 on the GPU; voxels sorted by key."""
        pts = _f64(points)
        out = np.empty_like(pts)
        n_out = C.c_int64(0)
        self._check(self._lib.icpmi_voxel_downsample(self._h, _dp(pts), pts.shape[0], float(voxel_size),
                                                     _dp(out), pts.shape[0], C.byref(n_out)))
        return out[:n_out.value].copy()

    def voxel_downsample_device(self, src_ptr, n, voxel_size, out_ptr, out_cap):
        n_out = C.c_int64(0)
        self._check(self._lib.icpmi_voxel_downsample_device(self._h, C.c_void_p(src_ptr), n, float(voxel_size),
                                                            C.c_void_p(out_ptr), out_cap, C.byref(n_out)))
        return n_out.value

    def estimate_normals_rows(self, points, k, row0, row1):
        """icp.hpp:23-67 for rows [row0, row1) only (against the whole cloud)"""
        pts = _f64(points)
        out = np.empty((row1 - row0, 3))
        self._check(self._lib.icpmi_estimate_normals_rows(self._h, _dp(pts), pts.shape[0], k, row0, row1, _dp(out)))
        return out

    def upload_points_f32(self, records, out_ptr):
        """records: n x stride float32 (x, y, z leading) -> n x 3 fp64 at device address out_ptr"""
        rec = np.ascontiguousarray(records, dtype=np.float32)
        self._check(self._lib.icpmi_upload_points_f32(self._h, rec.ctypes.data_as(C.POINTER(C.c_float)), rec.shape[0],
                                                      rec.shape[1], C.c_void_p(out_ptr)))

    def load_cloud_device_rows(self, path):
        n = C.c_int64(0)
        self._check(self._lib.icpmi_load_cloud_device(self._h, os.fsencode(path), None, 0, C.byref(n)))
        return n.value

    def load_cloud_device(self, path, out_ptr, cap):
        """load_ply / load_bin (file_utils.cpp:20-141) straight into device memory; returns the row count"""
        n = C.c_int64(0)
        self._check(self._lib.icpmi_load_cloud_device(self._h, os.fsencode(path), C.c_void_p(out_ptr), cap, C.byref(n)))
        return n.value

    def stream_push(self, raw_ptr, n_raw, voxel, min_points, cfg):
        """slam_node.cpp:122-152 on resident clouds -> (Result, error history, StreamInfo)"""
        cap = cfg.max_iterations + 1
        hist = np.zeros(max(cap, 1))
        res, info = Result(), StreamInfo()
        self._check(self._lib.icpmi_stream_push(self._h, C.c_void_p(raw_ptr), n_raw, float(voxel), int(min_points),
                                                C.byref(cfg), C.byref(res), _dp(hist), cap, C.byref(info)))
        return res, hist[:res.history_len].copy(), info

    def stream_push_host(self, raw, voxel, min_points, cfg):
        """stream_push with the raw scan in host memory (N x 3 fp64)"""
        pts = _f64(raw)
        cap = cfg.max_iterations + 1
        hist = np.zeros(max(cap, 1))
        res, info = Result(), StreamInfo()
        self._check(self._lib.icpmi_stream_push_host(self._h, _dp(pts), pts.shape[0], float(voxel), int(min_points),
                                                     C.byref(cfg), C.byref(res), _dp(hist), cap, C.byref(info)))
        return res, hist[:res.history_len].copy(), info

    def stream_push_file(self, path, voxel, min_points, cfg):
        """stream_push with the raw scan in a file (.bin: disk -> pinned memory -> device as float32)"""
        cap = cfg.max_iterations + 1
        hist = np.zeros(max(cap, 1))
        res, info = Result(), StreamInfo()
        self._check(self._lib.icpmi_stream_push_file(self._h, os.fsencode(path), float(voxel), int(min_points),
                                                     C.byref(cfg), C.byref(res), _dp(hist), cap, C.byref(info)))
        return res, hist[:res.history_len].copy(), info

    def stream_prefetch_file(self, path):
        """start reading the NEXT frame file on the context's worker thread (call before pushing the current one)"""
        self._check(self._lib.icpmi_stream_prefetch_file(self._h, os.fsencode(path)))

    def stream_reset(self):
        self._check(self._lib.icpmi_stream_reset(self._h))

    @staticmethod
    def make_grid_config(resolution=None, height_min=None, height_max=None, max_range=None):
        g = GridConfig()
        load_library().icpmi_grid_config_default(C.byref(g))
        for name, v in (("resolution", resolution), ("height_min", height_min), ("height_max", height_max),
                        ("max_range", max_range)):
            if v is not None:
                setattr(g, name, float(v))
        return g

    def occupancy_update(self, world, sensor, grid=None):
        """update_occupancy_grid(world, sensor) (slam_node.cpp:211-221) into the context's cell set -> its size"""
        pts = _f64(world) if len(world) else np.zeros((0, 3))
        sx = np.ascontiguousarray(sensor, dtype=np.float64).reshape(3)
        g = grid if grid is not None else self.make_grid_config()
        n = C.c_int64(0)
        self._check(self._lib.icpmi_occupancy_update(self._h, _dp(pts), pts.shape[0], _dp(sx), C.byref(g), C.byref(n)))
        return n.value

    def occupancy_update_device(self, ptr, n_rows, sensor, grid=None):
        sx = np.ascontiguousarray(sensor, dtype=np.float64).reshape(3)
        g = grid if grid is not None else self.make_grid_config()
        n = C.c_int64(0)
        self._check(self._lib.icpmi_occupancy_update_device(self._h, C.c_void_p(ptr), n_rows, _dp(sx), C.byref(g), C.byref(n)))
        return n.value

    def occupancy_cells(self):
        """the set as an (n, 2) int32 array of (x, y), sorted by x then y"""
        n = C.c_int64(0)
        self._check(self._lib.icpmi_occupancy_cells(self._h, None, 0, C.byref(n)))
        out = np.zeros((n.value, 2), dtype=np.int32)
        if n.value:
            self._check(self._lib.icpmi_occupancy_cells(self._h, out.ctypes.data_as(C.POINTER(C.c_int32)), n.value, C.byref(n)))
        return out

    def occupancy_clear(self):
        self._check(self._lib.icpmi_occupancy_clear(self._h))

    def stream_current_scan(self):
        """the filtered scan the last stream_push* left resident (slam_node.cpp:122's `curr`), N x 3 fp64"""
        n = C.c_int64(0)
        self._check(self._lib.icpmi_stream_current_scan(self._h, None, 0, C.byref(n)))
        out = np.empty((n.value, 3))
        if n.value:
            self._check(self._lib.icpmi_stream_current_scan(self._h, _dp(out), n.value, C.byref(n)))
        return out

    def stream_map_update(self, pose, grid=None, want_world=True, update_grid=True, n_rows=None):
        """slam_node.cpp:147-153 on the scan the last stream_push* left resident: (world points or None, cells in the set).
        n_rows: the filtered scan's row count (info.n_filtered of the push), which saves asking the library for it."""
        T = np.ascontiguousarray(pose, dtype=np.float64).reshape(16)
        nw, nc = C.c_int64(0), C.c_int64(0)
        g = (grid if grid is not None else self.make_grid_config()) if update_grid else None
        world = None
        if want_world:
            if n_rows is None:
                self._check(self._lib.icpmi_stream_map_update(self._h, _dp(T), None, None, 0, C.byref(nw), C.byref(nc)))
                n_rows = nw.value
            world = np.empty((int(n_rows), 3))
        self._check(self._lib.icpmi_stream_map_update(self._h, _dp(T), C.byref(g) if g is not None else None,
                                                      _dp(world) if world is not None else None,
                                                      world.shape[0] if world is not None else 0, C.byref(nw), C.byref(nc)))
        return world, nc.value

    def scan_context(self, cloud):
        """scan_context.hpp:44-82 -> 20 x 60 descriptor"""
        pts = _f64(cloud)
        out = np.empty(1200)
        self._check(self._lib.icpmi_scan_context(self._h, _dp(pts), pts.shape[0], _dp(out)))
        return out.reshape(20, 60)

    def scan_context_distances(self, query_desc, hist_descs):
        """scan_context.hpp:90-142 for one query against a stack of descriptors"""
        q = np.ascontiguousarray(query_desc, dtype=np.float64).reshape(1200)
        h = np.ascontiguousarray(hist_descs, dtype=np.float64).reshape(-1, 1200)
        out = np.empty(h.shape[0])
        self._check(self._lib.icpmi_scan_context_distances(self._h, _dp(q), _dp(h), h.shape[0], _dp(out)))
        return out

    # multi-GPU
    def comm_unique_id(self):
        buf = C.create_string_buffer(UNIQUE_ID_BYTES)
        self._check(self._lib.icpmi_comm_unique_id(self._h, buf))
        return buf.raw

    def comm_init(self, n_ranks, rank, unique_id):
        buf = C.create_string_buffer(bytes(unique_id), UNIQUE_ID_BYTES) if unique_id else None
        self._check(self._lib.icpmi_comm_init(self._h, n_ranks, rank, buf))
        self.n_ranks, self.rank = n_ranks, rank

    def comm_init_callbacks(self, n_ranks, rank, allreduce, allgather):
        """allreduce(np_array) / allgather(np_array, per_rank): in-place on a host numpy view."""
        def _ar(_user, buf, count):
            try:
                allreduce(np.ctypeslib.as_array(buf, shape=(count,)))
                return 0
            except Exception:  # noqa: BLE001 -- must not unwind through C
                import traceback
                traceback.print_exc()
                return 1

        def _ag(_user, buf, per):
            try:
                allgather(np.ctypeslib.as_array(buf, shape=(per * n_ranks,)), per)
                return 0
            except Exception:  # noqa: BLE001
                import traceback
                traceback.print_exc()
                return 1

        self._cb = (ALLREDUCE_FN(_ar), ALLGATHER_FN(_ag))  # keep alive
        self._check(self._lib.icpmi_comm_init_callbacks(self._h, n_ranks, rank, self._cb[0],
                                                        self._cb[1], None))
        self.n_ranks, self.rank = n_ranks, rank

    def comm_finalize(self):
        self._check(self._lib.icpmi_comm_finalize(self._h))
        self.n_ranks, self.rank = 1, 0

    def comm_info(self):
        """icpmi_comm_info (a collective): the communicator's own size and rank, every rank's device and PCI bus id."""
        ci = CommInfo()
        self._check(self._lib.icpmi_comm_info(self._h, C.byref(ci)))
        n = ci.n_ranks
        return {"kind": {0: "none", 1: "rccl", 2: "callbacks"}[ci.kind], "n_ranks": n, "rank": ci.rank,
                "devices": [int(ci.device[r]) for r in range(n)],
                "pci_bus_ids": [bytes(ci.pci_bus_id[r]).split(b"\0")[0].decode() for r in range(n)]}

    def reset_profile(self):
        self._check(self._lib.icpmi_reset_profile(self._h))

    def get_profile(self):
        p = Profile()
        self._check(self._lib.icpmi_get_profile(self._h, C.byref(p)))
        return {f[0]: getattr(p, f[0]) for f in Profile._fields_}
