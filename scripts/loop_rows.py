"""Diagnostic: the per-row matches the ICP loop's last pass left behind, bounded form (nn_bounded.h) against unbounded
form, for one seed of scripts/fuzz_bounded.py -- a diagnostic build of the library (-DICPMI_DEBUG_LOOP, /tmp) exports
them.  Every row whose match differs is printed with both distances and the brute-force truth.
    python scripts/loop_rows.py <seed> [iterations]"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
CSRC = os.path.join(ROOT, "lidar_slam_from_scratch_amd", "csrc")
so = "/tmp/libicp_dbg.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                       "-DICPMI_DEBUG_LOOP", "-c", "-o", "/tmp/capi_dbg.o", os.path.join(CSRC, "capi.hip")])
if not os.path.exists(os.path.join(CSRC, "sort.o")):
    subprocess.check_call(["make", "-s", "-C", CSRC, "sort.o"])
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, "/tmp/capi_dbg.o",
                       os.path.join(CSRC, "sort.o"), "-ldl"])
import numpy as np
import torch  # noqa: F401
from lidar_slam_from_scratch_amd import capi
L = capi.load_library(so)
L.icpmi_debug_loop_rows.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_uint32), C.c_int64]
from fuzz_bounded import make_case as make

seed = int(sys.argv[1])
src, tgt, iters, tol, info = make(seed)
if len(sys.argv) > 2:
    iters = int(sys.argv[2])
print(seed, info, "iterations", iters)
n = src.shape[0]
out = {}
for knob in ("1", "0"):
    os.environ["ICPMI_NN_BOUNDED"] = knob
    ctx = capi.Context(device=0, search=capi.SEARCH_MFMA_BF16)
    res, hist = ctx.align(src, tgt, capi.Context.make_config(iters, 0.0, 0.0))
    idx = np.empty(n, np.int32); cur = np.empty((n, 3)); perm = np.empty(n, np.uint32)
    rc = L.icpmi_debug_loop_rows(ctx._h, idx.ctypes.data_as(C.POINTER(C.c_int32)), cur.ctypes.data_as(C.POINTER(C.c_double)),
                                 perm.ctypes.data_as(C.POINTER(C.c_uint32)), n)
    assert rc == 0, rc
    if knob == "1":   # the lists the last bounded pass read
        L.icpmi_debug_loop_lists.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.c_int64]
        ub = np.empty(n); cnt = np.empty(n, np.int32); ent = np.empty((n, 8), np.uint32)
        if L.icpmi_debug_loop_lists(ctx._h, ub.ctypes.data_as(C.POINTER(C.c_double)), cnt.ctypes.data_as(C.POINTER(C.c_int32)),
                                    ent.ctypes.data_as(C.POINTER(C.c_uint32)), n) == 0:
            fin = np.isfinite(cur).all(1)
            print("  lists: rows", n, "finite", int(fin.sum()), "cnt==0 among finite", int(((cnt == 0) & fin).sum()), "cnt>8", int((cnt > 8).sum()),
                  "max |coordinate|", float(np.abs(cur[fin]).max()) if fin.any() else None, "ub range", float(np.nanmin(ub)), float(np.nanmax(ub)))
            z = np.nonzero((cnt == 0) & fin)[0][:5]
            for r in z:
                print("   empty list: row", r, "p", cur[r], "ub", ub[r], "prev/now match", idx[r])
    out[knob] = (idx, cur, perm, hist)
    print("knob", knob, "history", [float("%.9g" % h) for h in hist])
    ctx.close()
(i1, c1, p1, _), (i0, c0, p0, _) = out["1"], out["0"]
print("rows moved identically:", bool(np.array_equal(c1, c0, equal_nan=True)), " same order:", bool((p1 == p0).all()))
diff = np.nonzero(i1 != i0)[0]
print(len(diff), "rows with different matches")
for r in diff[:12]:
    p = c1[r]
    d = ((tgt - p) ** 2).sum(1)
    best = d.min(); where = np.nonzero(d == best)[0]
    print(" row", r, "src row", p1[r], "p", p, "bounded ->", i1[r], d[i1[r]] if i1[r] >= 0 else None, " unbounded ->", i0[r],
          d[i0[r]] if i0[r] >= 0 else None, " truth: d", best, "lowest index", where[0], "(%d tied)" % len(where))
