#!/usr/bin/env python3
"""Randomised cross-check of every search engine against the oracle's brute force (run on the GPU
box).  Sizes from 1 point up, awkward distributions (clusters, planes, lines, duplicates, far
offsets, mixed scales, a few non-finite rows).  Any difference in an index, a squared distance, a
neighbour list or a normal is printed with the seed that reproduces it; exit code 1 then.
usage: python scripts/fuzz_engines.py [trials] [first_seed]"""
import os, sys, time
import numpy as np
import torch  # noqa: F401
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lidar_slam_from_scratch_amd import capi
from oracle import oracle as orc

def cloud(rng, n, kind, scale, offset):
    if kind == "uniform":
        p = rng.uniform(-1, 1, (n, 3))
    elif kind == "clusters":
        c = rng.uniform(-1, 1, (max(1, n // 200 + 1), 3))
        p = c[rng.integers(0, c.shape[0], n)] + rng.normal(0, 1e-3, (n, 3))
    elif kind == "plane":
        p = rng.uniform(-1, 1, (n, 3)); p[:, 2] = 0.3 * p[:, 0] - 0.1 * p[:, 1] + rng.normal(0, 1e-4, n)
    elif kind == "line":
        t = rng.uniform(-1, 1, n); p = np.stack([t, 0.5 * t, -0.25 * t], axis=1) + rng.normal(0, 1e-5, (n, 3))
    elif kind == "grid":
        p = rng.integers(-6, 7, (n, 3)).astype(float) / 6.0          # many exact ties and duplicates
    else:  # mixed scales
        p = rng.uniform(-1, 1, (n, 3)) * (10.0 ** rng.integers(-3, 2, (n, 1)))
    return np.ascontiguousarray(p * scale + offset)

def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    ctxs = {"auto": capi.Context(device=0, search=capi.SEARCH_AUTO), "exact": capi.Context(device=0, search=capi.SEARCH_EXACT_F64),
            "mfma": capi.Context(device=0, search=capi.SEARCH_MFMA_BF16), "pruned": capi.Context(device=0, search=capi.SEARCH_MFMA_PRUNED)}
    kinds = ["uniform", "clusters", "plane", "line", "grid", "mixed"]
    bad = 0
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    stop_stats = {"decided": 0, "under": 0, "exhausted": 0, "min_margin": float("inf")}
    t0 = time.time()
    for t in range(trials):
        seed = seed0 + t
        rng = np.random.default_rng(seed)
        n_t = int(rng.choice([1, 2, 3, 7, 63, 64, 65, 255, 256, 257, 1000, 2047, 2048, 2049, 4100, 9000, 13000]))
        n_q = int(rng.choice([1, 5, 63, 64, 65, 300, 1500, 5000]))
        scale = float(10.0 ** rng.integers(-2, 3)); offset = rng.uniform(-1, 1, 3) * float(rng.choice([0.0, 1.0, 1e3, 1e5]))
        tkind = str(rng.choice(kinds))
        tgt = cloud(rng, n_t, tkind, scale, offset)
        qry = cloud(rng, n_q, str(rng.choice(kinds)), scale, offset)
        if rng.random() < 0.3 and n_t > 3:
            qry[: min(n_q, n_t) // 2] = tgt[rng.choice(n_t, min(n_q, n_t) // 2, replace=False)]   # queries that ARE targets
        if rng.random() < 0.15 and n_t > 10:
            tgt[rng.integers(0, n_t)] = np.nan; tgt[rng.integers(0, n_t), 1] = np.inf
        if rng.random() < 0.1:
            qry[rng.integers(0, n_q)] = np.nan
        oi, od = orc.nearest_batch_brute(tgt, qry)
        for name, ctx in ctxs.items():
            gi, gd = ctx.nearest_batch(tgt, qry)
            if not ((gi == oi).all() and ((gd == od) | (oi < 0)).all()):
                bad += 1
                print("MISMATCH nearest_batch seed %d engine %s n_t %d n_q %d: %d indices differ" % (seed, name, n_t, n_q, int((gi != oi).sum())))
        finite_t = np.isfinite(tgt).all(axis=1).all()
        if finite_t and n_t >= 3:
            k = int(rng.choice([1, 3, 20, 33]))
            for name in ("auto", "mfma"):
                gi, _ = ctxs[name].k_nearest(tgt, qry[:40], k)
                for r in range(min(40, n_q)):
                    if not np.isfinite(qry[r]).all():
                        continue
                    want = orc.k_nearest_brute(tgt, qry[r], k)
                    got = [j for j in gi[r] if j >= 0]
                    if got != list(want):
                        bad += 1
                        print("MISMATCH k_nearest seed %d engine %s n_t %d k %d row %d" % (seed, name, n_t, k, r))
                        break
            # normals: the oracle takes its neighbour lists from the kd-tree, whose choice among EQUAL
            # distances depends on the traversal (kdtree.hpp:160, strict <), the GPU's on the index: on
            # clouds with exact ties at the k-th neighbour the two sets differ by construction (the
            # k_nearest check above uses the brute-force oracle, which has the GPU's rule)
            if n_t <= 9000 and tkind != "grid":
                want = orc.estimate_normals(tgt, None, 20, nthreads=8)
                for name in ("auto", "mfma", "pruned"):
                    got = ctxs[name].estimate_normals(tgt, 20)
                    if not (got == want).all():
                        bad += 1
                        print("MISMATCH normals seed %d engine %s kind %s n_t %d: %d rows differ" % (seed, name, tkind, n_t, int((got != want).any(axis=1).sum())))
        # voxel filter: centroids bit-identical to the oracle's (sums in input order)
        if finite_t and n_t >= 1:
            vox = float(rng.choice([0.05, 0.2, 0.5])) * scale
            ext = np.abs(tgt - tgt.min(axis=0)).max() / vox
            if ext < 1.5e6:
                gv, ov = ctxs["auto"].voxel_downsample(tgt, vox), orc.voxel_downsample(tgt, vox)
                if gv.shape != ov.shape or not (gv == ov).all():
                    bad += 1
                    print("MISMATCH voxel seed %d n_t %d voxel %g" % (seed, n_t, vox))
        # a short registration on well-conditioned geometry: same error history as the oracle loop
        if finite_t and tkind in ("uniform", "clusters", "mixed") and n_t >= 300 and abs(offset).max() <= 1e3:
            from lidar_slam_from_scratch_amd import synth
            # a small motion about the cloud's own centre (one about a far-away origin throws the cloud
            # out of the basin of convergence, and a diverging ICP amplifies the last bit of every sum)
            T = synth.make_transform(rng.normal(0, 0.003, 3), rng.normal(0, 0.005, 3) * scale)
            c = tgt.mean(axis=0)
            src = np.ascontiguousarray((tgt[rng.permutation(n_t)[: max(64, n_t // 2)]] - c - T[:3, 3]) @ T[:3, :3] + c)
            ref = orc.icp_point_to_plane(src, tgt, 4, 0.0, 0.0, nthreads=8)
            for name in ("auto", "mfma", "pruned"):
                res, hist = ctxs[name].align(src, tgt, capi.Context.make_config(4, 0.0, 0.0))
                h = np.asarray(ref.error_history)
                if h[-1] > 2.0 * h[0] or (np.diff(h[1:]) > 0).any() or (len(h) > 1 and h[1] > 2.0 * h[0]):
                    continue   # (diverged or erratic -- also a FIRST step that makes the error much worse, the mark of a
                               #  near-singular system: every rounding difference is then amplified step by step; seed
                               #  131167, clusters, went 1.4e-5 -> 2.1e-4 in the oracle itself)
                # (clustered clouds give normal equations with condition numbers of 1e8 and more: the order of
                # the 28 sums then shows at 1e-6 relative in the history -- the first entry, which no solve
                # precedes, agrees to the last digit; the north_star's tolerance on the pose is 1e-4.  The source
                # is an exact subset of the target here, so the loop converges to rounding level: entries
                # under 1e-6 of the cloud's size are noise on both sides)
                dt, dr = synth.pose_delta(np.array(res.transformation[:]).reshape(4, 4), ref.transformation)
                if (res.num_iterations != ref.num_iterations or abs(hist[0] - ref.error_history[0]) > 1e-12 * max(1.0, scale)
                        or not np.allclose(hist, ref.error_history, rtol=1e-4, atol=1e-6 * scale) or dt > 1e-4 * scale or dr > 1e-4):
                    bad += 1
                    print("MISMATCH align seed %d engine %s kind %s n_t %d: %s vs %s" % (seed, name, tkind, n_t, hist, ref.error_history))
        # the stopping tests with the caller's settings (50, 1e-6, 1e-9) on well-conditioned pairs: iteration count,
        # flag and history length equal to the oracle's wherever the margin allows (scripts/fuzz_stopping.py), every
        # eighth trial (a registration to convergence costs what a dozen of the legs above do)
        if t % 8 == 7:
            import fuzz_stopping
            bad += fuzz_stopping.check_case(seed, {k: ctxs[k] for k in ("auto", "mfma", "pruned")}, fuzz_stopping.caller_config(), stop_stats)
        if (t + 1) % 20 == 0:
            print("%d trials, %d mismatches, %.0f s" % (t + 1, bad, time.time() - t0), flush=True)
    print("fuzz: %d trials, %d mismatches; stopping tests: %d registrations decided (smallest margin %.3g), %d under the 1e-12 floor"
          % (trials, bad, stop_stats["decided"], stop_stats["min_margin"], stop_stats["under"]))
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main())
