import os, sys, time, json
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from lidar_slam_from_scratch_amd import capi, synth
src, tgt, _ = synth.c3_uniform(100000)
ctx = capi.Context(device=0, search=capi.SEARCH_MFMA_BF16)
out = {}
for name, q in (("100k_queries", src), ("10k_queries", src[:10000])):
    ctx.k_nearest(tgt, q, 20)
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); idx, d2 = ctx.k_nearest(tgt, q, 20); ts.append((time.perf_counter() - t0) * 1e3)
    out[name] = {"call_ms_median": sorted(ts)[3], "checksum": int(idx.astype(np.int64).sum())}
ts = []
for _ in range(7):
    t0 = time.perf_counter(); n = ctx.estimate_normals_rows(tgt, 20, 25000, 50000); ts.append((time.perf_counter() - t0) * 1e3)
out["normals_rows_25k_of_100k"] = {"call_ms_median": sorted(ts)[3], "checksum": float(np.abs(n).sum())}
print(json.dumps(out))
