import sys, time, json
sys.path.insert(0, '.')
import numpy as np
from lidar_slam_from_scratch_amd import capi, synth
from oracle import oracle as orc
ctx = capi.Context(device=0, profile=True)
src, tgt, T = synth.c1_room_corner()
t0=time.time(); idx, d2 = ctx.nearest_batch(tgt, src); print("nn time", time.time()-t0)
oidx, od2 = orc.KDTree(tgt).nearest_batch(src)
print("nn idx equal", (idx==oidx).all(), "d2 bit-equal", (d2==od2).all())
t0=time.time(); nrm = ctx.estimate_normals(tgt, 20); print("normals time", time.time()-t0)
onrm = orc.estimate_normals(tgt, None, 20)
print("normals bit-equal frac", (nrm==onrm).all(axis=1).mean(), "max abs diff", np.abs(nrm-onrm).max())
cfg = capi.Context.make_config()
res, hist = ctx.align(src, tgt, cfg)
ref = orc.icp_point_to_plane(src, tgt)
print("gpu", res.converged, res.num_iterations, res.final_error, hist)
print("cpu", ref.converged, ref.num_iterations, ref.final_error, ref.error_history)
print("pose delta", synth.pose_delta(np.array(res.transformation[:]).reshape(4,4), ref.transformation))
# solve
mt = tgt[oidx]; mn = onrm[oidx]
Tg = ctx.solve_point_to_plane(src, mt, mn); Tc = orc.solve_point_to_plane(src, mt, mn)
print("solve maxdiff", np.abs(Tg-Tc).max())
# C3 timing
src3, tgt3, T3 = synth.c3_uniform()
cfg3 = capi.Context.make_config(max_iterations=30, tolerance=0.0, min_error=0.0)
for rep in range(2):
    ctx.reset_profile()
    t0=time.time(); res3, hist3 = ctx.align(src3, tgt3, cfg3); dt=time.time()-t0
    print("C3 call", dt, "s; iters", res3.num_iterations, "prof", ctx.get_profile())
print(hist3[:5], hist3[-3:])
t0=time.time(); ref3 = orc.icp_point_to_plane(src3, tgt3, 3, 0.0, 0.0, nthreads=1); print("cpu 3 iters", time.time()-t0, ref3.setup_seconds, ref3.loop_seconds, ref3.final_seconds)
print("hist cmp", ref3.error_history, hist3[:4])
