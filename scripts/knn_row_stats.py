"""Diagnostic: what k_knn_resolve does per row of a LiDAR-like frame -- slots listed for scanning, candidates under
the bound, collection attempts.  Builds a diagnostic library (-DICPMI_KNN_STOP=4: the kernel writes these three
numbers in place of a row's neighbour list) into /tmp; the product library is not touched.
usage (GPU box): python scripts/knn_row_stats.py"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "lidar_slam_from_scratch_amd", "csrc")
so = "/tmp/libicp_knnstat.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                       "-DICPMI_KNN_STOP=4", "-c", "-o", "/tmp/capi_knnstat.o", os.path.join(CSRC, "capi.hip")])
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, "/tmp/capi_knnstat.o",
                       os.path.join(CSRC, "sort.o"), "-ldl"])
import numpy as np
import torch  # noqa: F401
from lidar_slam_from_scratch_amd import capi, synth
capi.load_library(so)
ctx = capi.Context(device=0)
out = {}
for name, cloud in (("lidar_frame_3", synth.lidar_frame(3, voxel=0.5, **synth.DRIVE_200)),
                    ("uniform_8k", synth.c3_uniform(8000)[1])):
    idx, _d2 = ctx.k_nearest(cloud, cloud, 20)
    nf, total, attempts = idx[:, 0], idx[:, 1], idx[:, 2]
    out[name] = {"rows": int(cloud.shape[0]),
                 "slots_listed": {"mean": float(nf.mean()), "p50": int(np.percentile(nf, 50)), "p99": int(np.percentile(nf, 99)), "max": int(nf.max())},
                 "candidates": {"mean": float(total.mean()), "p50": int(np.percentile(total, 50)), "p99": int(np.percentile(total, 99)), "max": int(total.max())},
                 "attempts": {str(a): int((attempts == a).sum()) for a in np.unique(attempts)}}
print(json.dumps(out, indent=1))
