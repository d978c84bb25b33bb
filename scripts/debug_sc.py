import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from lidar_slam_from_scratch_amd import capi, synth
from oracle import oracle as orc
ctx = capi.Context(device=0)
for f in (0, 1, 7):
    c = synth.lidar_frame(f, beams=32, azimuths=900)
    g = ctx.scan_context(c); o = orc.scan_context(c)
    bad = np.argwhere(g != o)
    print("frame", f, "pts", c.shape[0], "diff bins", len(bad))
    for r, s in bad[:6]:
        print("   bin", r, s, "gpu", g[r, s], "cpu", o[r, s])
    # which points fall in a differing bin according to numpy
    if len(bad):
        x, y, z = c.T
        rng = np.sqrt(x * x + y * y); ang = np.arctan2(y, x) + np.pi
        ri = np.clip((rng / 4.0).astype(int), 0, 19); si = np.clip((ang / (2 * np.pi / 60)).astype(int), 0, 59)
        r, s = bad[0]
        m = (ri == r) & (si == s)
        print("   numpy members z:", np.sort(z[m])[-3:], "count", m.sum(), "frac angle", (ang[m] / (2 * np.pi / 60) % 1)[:5])
