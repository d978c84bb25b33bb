"""Where k_icp_small's time goes: a diagnostic build (-DICPMI_SMALL_CLOCKS: s_memtime stamps at the kernel's
phases, waves 0 and 7 of every workgroup, outstanding memory operations drained before each stamp) runs a
registration of two ~8k-point filtered frames with forced iterations; prints, per phase, the median over
workgroups of the cycles since the previous stamp, for the LAST pass.  Builds /tmp/libicp_sclk.so itself.
    python scripts/small_clock.py [points] [extra -D flags]"""
import ctypes as C, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "lidar_slam_from_scratch_amd", "csrc")
keep = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
extra = sys.argv[2:]
so = "/tmp/libicp_sclk.so"
flags = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-DICPMI_SMALL_CLOCKS"] + extra
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950"] + flags + ["-c", "-o", "/tmp/capi_sclk.o", os.path.join(CSRC, "capi.hip")])
if not os.path.exists(os.path.join(CSRC, "sort.o")):
    subprocess.check_call(["make", "-s", "-C", CSRC, "sort.o"])
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, "/tmp/capi_sclk.o", os.path.join(CSRC, "sort.o"), "-ldl"])
import numpy as np
import torch  # noqa: F401
from lidar_slam_from_scratch_amd import capi, synth
L = capi.load_library(so)
L.icpmi_debug_coarse_clocks.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int64]
rng = np.random.default_rng(7)
A = synth.lidar_frame(3, voxel=0.4, **synth.DRIVE_200)
B = synth.lidar_frame(4, voxel=0.4, **synth.DRIVE_200)
a = np.ascontiguousarray(A[np.sort(rng.choice(A.shape[0], min(keep, A.shape[0]), replace=False))])
b = np.ascontiguousarray(B[np.sort(rng.choice(B.shape[0], min(keep, B.shape[0]), replace=False))])
ctx = capi.Context(device=0)
cfg = capi.Context.make_config(30, 0.0, 0.0)
for _ in range(5):
    ctx.align(b, a, cfg)
nb, NS = (b.shape[0] + 31) // 32, 12
buf = (C.c_uint64 * (2 * NS * nb))()
assert L.icpmi_debug_coarse_clocks(ctx._h, buf, 2 * NS * nb) == 0
s = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 2, NS).astype(np.int64)
names = ["entry", "state, rows, operands, A", "matrix loop + records", "barrier", "selection + slot scans", "certificate", "terms", "sums stored"]
out = {"points": [int(b.shape[0]), int(a.shape[0])], "workgroups": nb, "phases_cycles_median": {}}
for w, tag in ((0, "wave0"), (1, "wave7")):
    d = {}
    for k in range(1, 8):
        d[names[k]] = float(np.median(s[:, w, k] - s[:, w, k - 1]))
    d["total"] = float(np.median(s[:, w, 7] - s[:, w, 0]))
    out["phases_cycles_median"][tag] = d
rt = s[:, :, 11]
out["kernel_span_us_first_entry_to_last_end"] = float((rt.max() - rt.min()) / 100.0)
print(json.dumps(out, indent=1))
