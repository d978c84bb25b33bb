"""Where a pass of the culled coarse kernel (k_nn_coarse_groups, nn_culled.h) spends its time on C3: a diagnostic build
(-DICPMI_GROUPS_CLOCKS: every workgroup stamps the 100 MHz clock at its entry, once its lists are known, at both ends of
each of its chunks and at its exit, into a buffer nothing else reads) runs one registration; the stamps of its LAST
pass are summarised: when workgroups start, how long their prologue is, how long a first / second chunk takes, when the
chip runs dry.  Run on the GPU box:
    python scripts/groups_clock.py [n] [iterations] [-D flags ...]
Builds /tmp/libicp_gclk.so itself; the product library is not touched."""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "lidar_slam_from_scratch_amd", "csrc")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 12
extra = sys.argv[3:]

so = "/tmp/libicp_gclk.so"
flags = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-DICPMI_GROUPS_CLOCKS"] + extra
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950"] + flags + ["-c", "-o", "/tmp/capi_gclk.o", os.path.join(CSRC, "capi.hip")])
if not os.path.exists(os.path.join(CSRC, "sort.o")):
    subprocess.check_call(["make", "-s", "-C", CSRC, "sort.o"])
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, "/tmp/capi_gclk.o",
                       os.path.join(CSRC, "sort.o"), "-ldl"])

import numpy as np
import torch
from lidar_slam_from_scratch_amd import capi, synth

L = capi.load_library(so)
L.icpmi_debug_coarse_clocks.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int64]
src, tgt, _ = synth.c3_uniform(n)
dsrc, dtgt = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()
cfg = capi.Context.make_config(iters, 0.0, 0.0)
ctx = capi.Context(device=0, search=0, profile=0)
for _ in range(2):
    res, hist = ctx.align_device(dsrc.data_ptr(), n, dtgt.data_ptr(), n, cfg)
waves = 4 if os.environ.get("ICPMI_GROUPS_WAVES", "8")[:1] == "4" else 8
per_cu = int(os.environ.get("ICPMI_GROUPS_GRID", "0")) or 16 // waves
wgs = per_cu * 256
K = 12
buf = (C.c_uint64 * (K * wgs))()
assert L.icpmi_debug_coarse_clocks(ctx._h, buf, K * wgs) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(wgs, K).astype(np.int64)
if os.environ.get("GROUPS_CLOCK_RAW"):
    np.save(os.environ["GROUPS_CLOCK_RAW"], a)
t0 = a[:, 0].min()
us = lambda x: (x - t0) / 100.0
nch = a[:, 11]
entry, ready, exit_ = us(a[:, 0]), us(a[:, 1]), us(a[:, 10])
out = {"points": n, "workgroups": wgs, "waves_per_workgroup": waves, "chunks_total": int(nch.sum()),
       "chunks_per_workgroup": {str(k): int((nch == k).sum()) for k in range(0, 5)},
       "entry_us": {"p50": float(np.median(entry)), "max": float(entry.max())},
       "prologue_us": {"p50": float(np.median(ready - entry)), "p90": float(np.percentile(ready - entry, 90))},
       "exit_us": {"p10": float(np.percentile(exit_, 10)), "p50": float(np.median(exit_)), "p90": float(np.percentile(exit_, 90)), "max": float(exit_.max())}}
for k in range(3):
    m = nch > k
    if m.any():
        d = (a[m, 3 + 2 * k] - a[m, 2 + 2 * k]) / 100.0
        gap = (a[m, 2 + 2 * k] - (a[m, 1] if k == 0 else a[m, 1 + 2 * k])) / 100.0
        out["chunk%d_us" % k] = {"n": int(m.sum()), "p50": float(np.median(d)), "p90": float(np.percentile(d, 90)),
                                  "start_p50": float(np.median(us(a[m, 2 + 2 * k]))), "gap_before_p50": float(np.median(gap))}
bins = np.arange(0.0, exit_.max() + 2.0, 2.0)
busy = []
for b in bins:
    c = 0
    for k in range(3):
        m = nch > k
        c += int(((us(a[m, 2 + 2 * k]) < b + 2.0) & (us(a[m, 3 + 2 * k]) > b)).sum())
    busy.append(c)
out["chunks_in_flight_per_2us_bin"] = busy
print(json.dumps(out))
