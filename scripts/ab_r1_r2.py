"""Round-1 library against the current one, same process, same box, interleaved calls (boxes of the
pool differ by +-5 %, more than a round's kernel work moves the C3 call).  scripts/ab_r1_libicp.so is
the library of commit 0fd41fd (round 1), built from `git archive` of its csrc/."""
import ctypes as C, json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lidar_slam_from_scratch_amd import capi, synth

def bind(path):
    L = C.CDLL(path)
    vp = C.c_void_p
    L.icpmi_create.argtypes = [C.POINTER(capi.Options), C.POINTER(vp)]
    L.icpmi_destroy.argtypes = [vp]; L.icpmi_destroy.restype = None
    L.icpmi_options_default.argtypes = [C.POINTER(capi.Options)]
    L.icpmi_align_device.argtypes = [vp, vp, C.c_int64, vp, C.c_int64, C.POINTER(capi.Config), C.POINTER(capi.Result),
                                     C.POINTER(C.c_double), C.c_int32]
    return L

def make(L, search=0):
    o = capi.Options(); L.icpmi_options_default(C.byref(o)); o.search = search; o.profile = 0
    h = C.c_void_p(); assert L.icpmi_create(C.byref(o), C.byref(h)) == 0
    return h

capi.load_library()
libs = {"r1": bind(os.path.join(ROOT, "scripts", "ab_r1_libicp.so")), "r2": bind(capi._build.LIB_PATH)}
src, tgt, _ = synth.c3_uniform(100_000)
dsrc, dtgt = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()
torch.cuda.synchronize()
out = {}
for search, name in ((0, "all_pairs"), (3, "pruned")):
    for iters in (20, 30):
        cfg = capi.Context.make_config(iters, 0.0, 0.0)
        hist = np.zeros(iters + 1); res = capi.Result()
        ctx = {k: make(L, search) for k, L in libs.items()}
        ts = {k: [] for k in libs}
        for rep in range(8):
            for k, L in libs.items():
                t0 = time.perf_counter()
                rc = L.icpmi_align_device(ctx[k], C.c_void_p(dsrc.data_ptr()), 100_000, C.c_void_p(dtgt.data_ptr()), 100_000,
                                          C.byref(cfg), C.byref(res), hist.ctypes.data_as(C.POINTER(C.c_double)), iters + 1)
                dt = time.perf_counter() - t0
                assert rc == 0
                if rep >= 2:
                    ts[k].append(dt)
        out["%s_%d_iterations" % (name, iters)] = {k: {"ms_per_call_median": round(1e3 * float(np.median(v)), 3),
                                                       "it_per_s": round(iters / float(np.median(v)), 1)} for k, v in ts.items()}
        for k, L in libs.items():
            L.icpmi_destroy(ctx[k])
print(json.dumps(out, indent=1))
