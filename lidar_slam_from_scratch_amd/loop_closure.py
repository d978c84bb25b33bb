"""Loop-closure detection: Scan Context candidates + ICP verification, the second caller of
the ICP hot path.  Host-side mirror of the reference's
    LoopClosureConfig / LoopClosureResult / LoopClosureDetector   (core/loop_closure.hpp)
    ScanContext                                                   (core/scan_context.hpp)
with the arithmetic behind a small backend object:
    backend.scan_context(cloud) -> (20, 60) descriptor            scan_context.hpp:44-82
    backend.distances(query_desc, hist_descs) -> array            scan_context.hpp:90-142
    backend.align(source, target, max_iterations, tolerance)      icp.hpp:157-258
`GpuBackend` goes through the C ABI; the parity tests plug the oracle in instead.
"""
import numpy as np

MAX_BATCH = 8   # ICPMI_MAX_BATCH (include/icp_mi355x.h): registrations one icpmi_align_batch call takes


class LoopClosureConfig:
    """loop_closure.hpp:14-19"""

    def __init__(self, frame_gap=50, sc_distance_threshold=0.25, icp_fitness_threshold=0.3, max_candidates=3):
        self.frame_gap = frame_gap
        self.sc_distance_threshold = sc_distance_threshold
        self.icp_fitness_threshold = icp_fitness_threshold
        self.max_candidates = max_candidates


class LoopClosureResult:
    """loop_closure.hpp:25-31"""

    def __init__(self, query_frame, match_frame, transform, scan_context_distance, icp_fitness):
        self.query_frame = query_frame
        self.match_frame = match_frame
        self.transform = transform
        self.scan_context_distance = scan_context_distance
        self.icp_fitness = icp_fitness


class GpuBackend:
    def __init__(self, ctx):
        self.ctx = ctx

    def scan_context(self, cloud):
        return self.ctx.scan_context(cloud)

    def distances(self, query_desc, hist_descs):
        return self.ctx.scan_context_distances(query_desc, hist_descs)

    def align(self, source, target, max_iterations, tolerance):
        from .odometry import gpu_align
        return gpu_align(self.ctx)(source, target, max_iterations, tolerance)

    def align_many(self, source, targets, max_iterations, tolerance):
        """The verifications of one detect() side by side on the GPU (icpmi_align_batch): same results as
        align() one after the other."""
        from . import capi

        class _R:
            pass

        cfg = capi.Context.make_config(max_iterations=max_iterations, tolerance=tolerance)
        out = []
        for res, _hist in self.ctx.align_batch([source] * len(targets), targets, cfg):
            r = _R()
            r.transformation = np.array(res.transformation[:]).reshape(4, 4)
            r.converged, r.final_error, r.num_iterations = bool(res.converged), res.final_error, res.num_iterations
            out.append(r)
        return out


class LoopClosureDetector:
    """loop_closure.hpp:41-148"""

    def __init__(self, backend, config=None):
        self.backend = backend
        self.config = config or LoopClosureConfig()
        self.clear()

    def add_frame(self, cloud, frame_idx):
        """loop_closure.hpp:54-60"""
        cloud = np.ascontiguousarray(cloud, dtype=np.float64)
        self._descriptors.append(np.asarray(self.backend.scan_context(cloud)).reshape(20, 60))
        self._clouds.append(cloud)
        self._frame_indices.append(int(frame_idx))
        self._latest = int(frame_idx)

    def size(self):
        return len(self._descriptors)

    def clear(self):
        self._descriptors, self._clouds, self._frame_indices, self._latest = [], [], [], -1

    def detect(self):
        """loop_closure.hpp:66-126: closures for the most recently added frame."""
        results = []
        if len(self._descriptors) < 2:
            return results
        q = len(self._descriptors) - 1
        hist = np.stack(self._descriptors[:-1])
        dist = self.backend.distances(self._descriptors[q], hist)            # :86 for every i
        candidates = []
        for i in range(q):
            if self._frame_indices[q] - self._frame_indices[i] < self.config.frame_gap:   # :81-82
                continue
            if dist[i] < self.config.sc_distance_threshold:                   # :87-89
                candidates.append((float(dist[i]), i))
        candidates.sort()                                                     # :93
        # The reference verifies the candidates one after the other until max_candidates are ACCEPTED (:96-123).
        # The registrations are independent, so the next (max_candidates - accepted) of them -- all of which the
        # sequential loop would reach -- run side by side when the backend can (icpmi_align_batch); the
        # outcomes are taken in the reference's order.
        verified, pos = 0, 0
        many = getattr(self.backend, "align_many", None)
        while pos < len(candidates) and verified < self.config.max_candidates:   # :97
            # (at most MAX_BATCH side by side: icpmi_align_batch's limit; a larger max_candidates takes more rounds)
            chunk = candidates[pos:pos + min(self.config.max_candidates - verified, MAX_BATCH)]
            pos += len(chunk)
            if many is not None and len(chunk) > 1:
                outs = many(self._clouds[q], [self._clouds[c] for _, c in chunk], 30, 1e-6)
            else:
                outs = [self.backend.align(self._clouds[q], self._clouds[c], 30, 1e-6) for _, c in chunk]   # :102-109
            for (sc_dist, cand), r in zip(chunk, outs):
                if r.converged and r.final_error < self.config.icp_fitness_threshold:   # :112
                    results.append(LoopClosureResult(self._frame_indices[q], self._frame_indices[cand],
                                                     np.asarray(r.transformation), sc_dist, r.final_error))
                    verified += 1
        return results
