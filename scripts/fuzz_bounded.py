#!/usr/bin/env python3
"""Randomised cross-check of the bounded ICP passes (nn_bounded.h) against the unbounded ones (run on the GPU box): the
same registration with ICPMI_NN_BOUNDED=1 and =0 on the all-pairs engine, and on the culled engine (bounded passes on
the (64-row group, split) pairs within reach, nn_culled.h), in one process must give the same bits -- pose, history,
iteration count -- whatever the clouds: sizes on both sides of the two resolve layouts and coarse unit sizes, awkward distributions
(clusters, planes, lines, exact ties and duplicates, mixed scales, far offsets), motions from a hair to far outside the
basin (every row then takes the exhaustive path), NaN / infinite rows.  Any difference is printed with the seed that
reproduces it; exit code 1 then.
usage: python scripts/fuzz_bounded.py [trials] [first_seed]"""
import os, sys, time
import numpy as np
import torch  # noqa: F401
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lidar_slam_from_scratch_amd import capi, synth
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from fuzz_engines import cloud


KINDS = ["uniform", "clusters", "plane", "line", "grid", "mixed"]


def make_case(seed):
    """The registration of trial `seed` (scripts/loop_rows.py rebuilds single cases from it): source, target, iterations,
    tolerance, a description.  Seeds from 40000 up also draw offsets of 1e5 cloud sizes."""
    rng = np.random.default_rng(seed)
    n_t = int(rng.choice([33000, 40000, 70000, 120000]))          # more than 16 splits: the general kernels
    n_s = int(rng.choice([4096, 5000, 12500, 32768, 32769, 50000]))
    scale = float(10.0 ** rng.integers(-2, 3))
    offset = rng.uniform(-1, 1, 3) * float(rng.choice([0.0, 1.0, 1e3] if seed < 40000 else [0.0, 1.0, 1e3, 1e5]))
    tk = str(rng.choice(KINDS))
    tgt = cloud(rng, n_t, tk, scale, offset)
    if rng.random() < 0.5:
        pick = rng.choice(n_t, min(n_s, n_t), replace=False)
        src = tgt[pick] + rng.normal(0, 1e-3 * scale, (pick.shape[0], 3))
        sk = "subset"
    else:
        sk = str(rng.choice(KINDS))
        src = cloud(rng, n_s, sk, scale, offset)
    motion = float(rng.choice([1e-4, 1e-2, 0.3, 3.0]))
    T = synth.make_transform(rng.normal(0, 0.05 * min(motion, 1.0), 3), rng.normal(0, motion, 3) * scale)
    c = tgt.mean(axis=0)
    src = np.ascontiguousarray((src - c) @ T[:3, :3].T + T[:3, 3] + c)
    nanrows = False
    if rng.random() < 0.2:
        src[rng.integers(0, src.shape[0])] = np.nan
        src[rng.integers(0, src.shape[0]), 2] = np.inf
        nanrows = True
    iters = int(rng.choice([2, 5, 9]))
    tol = 0.0 if rng.random() < 0.5 else 1e-6
    return src, tgt, iters, tol, dict(target=tk, source=sk, scale=scale, motion=motion, nanrows=nanrows, n_t=n_t, n_s=src.shape[0])


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    ctxs = {"all pairs": capi.Context(device=0, search=capi.SEARCH_MFMA_BF16, profile=2),
            "pruned": capi.Context(device=0, search=capi.SEARCH_MFMA_PRUNED, profile=2)}
    bad, bounded_total, exhaustive_total = 0, 0, 0
    t0 = time.time()
    for t in range(trials):
        seed = seed0 + t
        src, tgt, iters, tol, info = make_case(seed)
        n_t = info["n_t"]
        cfg = capi.Context.make_config(iters, tol, 0.0)
        # the reference: the all-pairs engine with every pass unbounded (coarse minima + certificate, round 2's form); against
        # it, bit for bit, the all-pairs engine's bounded passes and the culled engine (which has bounded passes only; the rows
        # are in the same Morton order in all three, so the sums are formed in the same order)
        out = {}
        for engine, knob in (("all pairs", "0"), ("all pairs", "1"), ("pruned", "1")):
            ctx = ctxs[engine]
            os.environ["ICPMI_NN_BOUNDED"] = knob
            ctx.reset_profile()
            res, hist = ctx.align(src, tgt, cfg)
            p = ctx.get_profile()
            out[(engine, knob)] = (tuple(res.transformation[:]), tuple(hist), res.num_iterations, bool(res.converged), res.final_error)
            if knob == "1":
                bounded_total += int(p["bounded_launches"]); exhaustive_total += int(p["nn_fallback_queries"])
        ref = out[("all pairs", "0")]
        for key in (("all pairs", "1"), ("pruned", "1")):
            got = out[key]
            same = all((a == b) or (isinstance(a, float) and a != a and b != b) for a, b in zip(got[2:], ref[2:])) and \
                np.array_equal(np.array(got[0]), np.array(ref[0]), equal_nan=True) and \
                np.array_equal(np.array(got[1]), np.array(ref[1]), equal_nan=True)
            if not same:
                bad += 1
                print("MISMATCH seed %d engine %s n_t %d n_s %d iters %d: %s vs %s" % (seed, key[0], n_t, src.shape[0], iters, got[1], ref[1]))
        if (t + 1) % 10 == 0:
            print("%d trials, %d mismatches, %d bounded passes, %d rows searched exhaustively, %.0f s"
                  % (t + 1, bad, bounded_total, exhaustive_total, time.time() - t0), flush=True)
    print("fuzz_bounded: %d trials, %d mismatches, %d bounded passes, %d rows searched exhaustively" % (trials, bad, bounded_total, exhaustive_total))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
