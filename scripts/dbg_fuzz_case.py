import os, sys
import numpy as np
import torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from lidar_slam_from_scratch_amd import capi, synth
from fuzz_engines import cloud

def make(seed):
    rng = np.random.default_rng(seed)
    kinds = ["uniform", "clusters", "plane", "line", "grid", "mixed"]
    n_t = int(rng.choice([33000, 40000, 70000, 120000])); n_s = int(rng.choice([4096, 5000, 12500, 32768, 32769, 50000]))
    scale = float(10.0 ** rng.integers(-2, 3)); offset = rng.uniform(-1, 1, 3) * float(rng.choice([0.0, 1.0, 1e3]))
    tk = str(rng.choice(kinds))
    tgt = cloud(rng, n_t, tk, scale, offset)
    if rng.random() < 0.5:
        pick = rng.choice(n_t, min(n_s, n_t), replace=False); src = tgt[pick] + rng.normal(0, 1e-3 * scale, (pick.shape[0], 3)); sk = "subset"
    else:
        sk = str(rng.choice(kinds)); src = cloud(rng, n_s, sk, scale, offset)
    motion = float(rng.choice([1e-4, 1e-2, 0.3, 3.0]))
    T = synth.make_transform(rng.normal(0, 0.05 * min(motion, 1.0), 3), rng.normal(0, motion, 3) * scale)
    c = tgt.mean(axis=0)
    src = np.ascontiguousarray((src - c) @ T[:3, :3].T + T[:3, 3] + c)
    nanrows = False
    if rng.random() < 0.2:
        src[rng.integers(0, src.shape[0])] = np.nan; src[rng.integers(0, src.shape[0]), 2] = np.inf; nanrows = True
    iters = int(rng.choice([2, 5, 9]))
    tol = 0.0 if rng.random() < 0.5 else 1e-6
    return src, tgt, iters, tol, dict(tk=tk, sk=sk, scale=scale, motion=motion, nanrows=nanrows, n_t=n_t, n_s=src.shape[0])

