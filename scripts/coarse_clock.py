"""The clock the chip holds INSIDE k_nn_coarse<0,..> on C3 (MI355X_MICROARCH.md, DVFS give-back item 6):
a diagnostic build of the library (-DICPMI_COARSE_CLOCKS: one workgroup-uniform pair of s_memtime /
s_memrealtime stamps at each end of a (query block, split) unit, written to a buffer nothing else
reads) runs C3 calls back to back for >= 2 s; clock = d(s_memtime) / d(s_memrealtime) x 100 MHz per
workgroup, median over the workgroups of the last pass; cycles per MFMA follow from the 128 MFMAs a
wave issues per unit at 4 waves per SIMD.  Run on the GPU box:
    python scripts/coarse_clock.py [n] [seconds]
Builds /tmp/libicp_clk.so itself; the product library is not touched."""
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "lidar_slam_from_scratch_amd", "csrc")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 2.5
extra = sys.argv[3:]  # further -D flags for A/B builds

so = "/tmp/libicp_clk.so"
flags = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-DICPMI_COARSE_CLOCKS"] + extra
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950"] + flags + ["-c", "-o", "/tmp/capi_clk.o",
                                                                                  os.path.join(CSRC, "capi.hip")])
if not os.path.exists(os.path.join(CSRC, "sort.o")):
    subprocess.check_call(["make", "-s", "-C", CSRC, "sort.o"])
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, "/tmp/capi_clk.o",
                       os.path.join(CSRC, "sort.o"), "-ldl"])

os.environ["ICPMI_NN_BOUNDED"] = "0"   # the stamps sit in the certified form's pass (k_nn_coarse<0>: the same unit and loop as the bounded
                                       # pass's); since round 4 no registration runs it unless asked
import numpy as np
import torch
from lidar_slam_from_scratch_amd import capi, synth

L = capi.load_library(so)
L.icpmi_debug_coarse_clocks.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int64]
src, tgt, _ = synth.c3_uniform(n)
dsrc, dtgt = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()
cfg = capi.Context.make_config(30, 0.0, 0.0)
ctx = capi.Context(device=0, search=2, profile=0)
t0 = time.perf_counter()
calls = 0
while time.perf_counter() - t0 < seconds:
    res, hist = ctx.align_device(dsrc.data_ptr(), n, dtgt.data_ptr(), n, cfg)
    calls += 1
wall = time.perf_counter() - t0
splits = (n + 2047) // 2048
units = ((n + 511) // 512) * splits
buf = (C.c_uint64 * (4 * units))()
rc = L.icpmi_debug_coarse_clocks(ctx._h, buf, 4 * units)
assert rc == 0, rc
a = np.frombuffer(buf, dtype=np.uint64).reshape(units, 4).astype(np.int64)
dc, dr = a[:, 2] - a[:, 0], a[:, 3] - a[:, 1]
ok = dr > 0
ghz = dc[ok] / dr[ok] * 0.1
span_us = (a[:, 3].max() - a[:, 1].min()) / 100.0
# the last pass as a timeline: units in flight per 10 us bin, and how long a unit takes by when it starts
t0s = (a[:, 1] - a[:, 1].min()) / 100.0
t1s = (a[:, 3] - a[:, 1].min()) / 100.0
bins = np.arange(0.0, t1s.max() + 10.0, 10.0)
in_flight = [int(((t0s < b + 10.0) & (t1s > b)).sum()) for b in bins]
order = np.argsort(t0s)
dec = np.array_split(order, 10)
unit_us_by_start_decile = [round(float(np.median(t1s[i] - t0s[i])), 2) for i in dec]
print(json.dumps({
    "points": n, "calls": calls, "ms_per_call": round(1e3 * wall / calls, 3), "units": int(units),
    "in_kernel_clock_GHz": {"median": round(float(np.median(ghz)), 3), "p10": round(float(np.percentile(ghz, 10)), 3),
                            "p90": round(float(np.percentile(ghz, 90)), 3)},
    "unit_us_median": round(float(np.median(dr[ok])) / 100.0, 2),
    "unit_cycles_median": int(np.median(dc[ok])),
    "cycles_per_mfma_at_4_waves_per_simd": round(float(np.median(dc[ok])) / (128 * 4), 1),
    "last_pass_first_stamp_to_last_us": round(float(span_us), 1),
    "units_in_flight_per_10us_bin": in_flight, "unit_us_by_start_decile": unit_us_by_start_decile,
    "extra_flags": extra,
}))
