#!/usr/bin/env python3
"""What "same iteration count" means against an unpinned oracle: the loop leaves on
|prev_error - error| < tolerance (icp.hpp:214), and the error is a sum of N squares whose last bits
depend on the ORDER of the additions -- Eigen's packetised sums, this oracle's index order and the GPU's
tree order all differ.  For a registration this script restates the oracle loop in Python on the oracle's
own primitives (kd-tree search, normals, index-order normal equations, pivoted LDLT) and records

  margin  = min over every evaluated iteration of | |prev - err| - tolerance |  and | err - min_error |
            (how far the nearest stopping test was from flipping), and
  spread  = max over the iterations of (max - min) of err under three summation orders of the N squared
            residuals: index order, reversed, NumPy's pairwise.

margin >> spread  <=>  no summation order can change the iteration count through the ERROR; what is left is
the order of the 27 normal-equation sums acting through the 6x6 solve, which kappa(J^T J) amplifies: the same
loop is run again with the rows fed in reversed order (`order="reversed"`) and the counts compared.

    python scripts/iteration_sensitivity.py            # fixtures + the 200-frame drive's 199 registrations + a kappa sweep
    python scripts/iteration_sensitivity.py --quick    # fixtures, 30 small frames, a short sweep (what the CPU test runs)
Test infrastructure only (imports oracle/)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def _apply(T, P):
    """P * R^T + t^T in the reference's operation order (icp.hpp:225-226)"""
    R, t = T[:3, :3], T[:3, 3]
    return np.stack([((P[:, 0] * R[a, 0] + P[:, 1] * R[a, 1]) + P[:, 2] * R[a, 2]) + t[a] for a in range(3)], axis=1)


def registration_margins(src, tgt, max_iterations=50, tolerance=1e-6, min_error=1e-9, order="index", nthreads=1):
    """The oracle loop (icp.hpp:157-258) with the stopping tests' margins and the error's spread under three
    summation orders.  order = "index" | "reversed": the order in which the rows enter the normal equations
    AND the error that steers the loop."""
    from oracle import oracle as orc
    src = np.ascontiguousarray(src, dtype=np.float64)
    tgt = np.ascontiguousarray(tgt, dtype=np.float64)
    tree = orc.KDTree(tgt)
    normals = orc.estimate_normals(tgt, tree, 20, nthreads=nthreads)
    cur, total = src.copy(), np.eye(4)
    prev = np.finfo(np.float64).max
    hist, margin, spread, kappa, conv = [], np.inf, 0.0, None, False
    rev = order == "reversed"

    def errors(cur):
        idx, _ = tree.nearest_batch(cur, nthreads=nthreads)
        q, nn = tgt[idx], normals[idx]
        d = q - cur
        b = (d[:, 0] * nn[:, 0] + d[:, 1] * nn[:, 1]) + d[:, 2] * nn[:, 2]     # icp.hpp:116
        sq = b * b
        n = float(cur.shape[0])
        e_index = float(np.sqrt(np.cumsum(sq)[-1] / n))                          # sequential, index order (icp.hpp:198-206)
        e_rev = float(np.sqrt(np.cumsum(sq[::-1])[-1] / n))
        e_pair = float(np.sqrt(np.sum(sq) / n))                                  # NumPy: pairwise blocks
        return idx, (e_index, e_rev, e_pair)

    for _ in range(max_iterations):
        idx, es = errors(cur)
        err = es[1] if rev else es[0]
        hist.append(err)
        spread = max(spread, max(es) - min(es))
        margin = min(margin, abs(abs(prev - err) - tolerance), abs(err - min_error))
        if err < min_error or abs(prev - err) < tolerance:                       # icp.hpp:210-217
            conv = True
            break
        rows = slice(None, None, -1) if rev else slice(None)
        sums = orc.normal_equations(np.ascontiguousarray(cur[rows]), np.ascontiguousarray(tgt[idx][rows]),
                                    np.ascontiguousarray(normals[idx][rows]))
        if kappa is None:
            A = np.zeros((6, 6))
            A[np.triu_indices(6)] = sums[:21]
            A = A + A.T - np.diag(np.diag(A))
            kappa = float(np.linalg.cond(A))
        delta = orc.solve_from_sums(sums)
        cur = _apply(delta, cur)
        total = delta @ total
        prev = err
    idx, es = errors(cur) if not conv else (None, (hist[-1],) * 3)
    hist.append(es[1] if rev else es[0])
    return {"num_iterations": len(hist) - 1, "converged": conv, "history": hist, "margin": float(margin), "spread": float(spread),
            "kappa": kappa, "transformation": total}


def near_degenerate_pair(theta, n=4000, seed=11):
    """A registration whose normal equations lose rank as theta -> 0: three planes over the same 20 m square whose
    normals are theta apart (z = 0, z = tan(theta) y + 10, z = tan(theta) x + 20: far enough apart that no 20-neighbourhood mixes two of them): translation along x and y and the
    rotation about z are observable through sin(theta) only, kappa(J^T J) ~ theta^-2."""
    rng = np.random.default_rng(seed)
    xy = rng.uniform(-10, 10, (3, n, 2))
    tt = np.tan(theta)
    planes = [np.column_stack([xy[0], np.zeros(n)]), np.column_stack([xy[1], tt * xy[1][:, 1] + 10.0]),
              np.column_stack([xy[2], tt * xy[2][:, 0] + 20.0])]
    noise = 1e-7  # (centimetre noise would make the estimated normals scatter by more than theta and hide the degeneracy)
    tgt = np.vstack(planes) + rng.normal(0, noise, (3 * n, 3))
    from lidar_slam_from_scratch_amd import synth
    T = synth.make_transform((0.004, -0.003, 0.005), (0.03, -0.02, 0.01))
    pick = rng.permutation(tgt.shape[0])[: 2 * n]
    src = synth.apply_transform(synth.invert_transform(T), tgt[pick] + rng.normal(0, noise, (2 * n, 3)))
    return src, tgt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    from lidar_slam_from_scratch_amd import synth
    from oracle import oracle as orc
    nth = os.cpu_count() or 1
    out = {"fixtures": {}, "stream": {}, "kappa_sweep": []}
    fixtures = {"c1_room_corner": synth.c1_room_corner()[:2], "c2_lidar_pair": synth.c2_lidar_pair()[:2]}
    for name, (s, t) in fixtures.items():
        r = registration_margins(s, t, nthreads=nth)
        ref = orc.icp_point_to_plane(s, t, nthreads=nth)
        out["fixtures"][name] = {"iterations": r["num_iterations"], "margin": r["margin"], "spread": r["spread"], "kappa": r["kappa"],
                                 "ratio": r["margin"] / max(r["spread"], 1e-300), "restatement_equals_oracle": bool(
                                     r["num_iterations"] == ref.num_iterations and np.array_equal(r["history"], ref.error_history))}
    frames, kw = (30, dict(beams=32, azimuths=900)) if args.quick else (200, dict())
    prev, rows = None, []
    for f in range(frames):
        cur = synth.lidar_frame(f, voxel=0.5, **kw, **synth.DRIVE_200)
        if prev is not None and cur.shape[0] >= 1000:
            r = registration_margins(cur, prev, nthreads=nth)
            r2 = registration_margins(cur, prev, order="reversed", nthreads=nth)
            rows.append({"frame": f, "iterations": r["num_iterations"], "iterations_reversed_order": r2["num_iterations"],
                         "margin": r["margin"], "spread": r["spread"], "kappa": r["kappa"]})
        prev = cur
    conv = [x for x in rows if x["iterations"] < 50]
    out["stream"] = {"registrations": len(rows), "ran_out_of_iterations": len(rows) - len(conv),
                     "min_margin": min(x["margin"] for x in rows), "max_spread": max(x["spread"] for x in rows),
                     "min_ratio": min(x["margin"] / max(x["spread"], 1e-300) for x in rows),
                     "max_kappa": max(x["kappa"] for x in rows),
                     "counts_equal_under_reversed_order": sum(x["iterations"] == x["iterations_reversed_order"] for x in rows),
                     "worst": sorted(rows, key=lambda x: x["margin"] / max(x["spread"], 1e-300))[:3]}
    thetas = [0.3, 0.03, 3e-3, 3e-4, 3e-5] if args.quick else [0.3, 0.1, 0.03, 0.01, 3e-3, 1e-3, 3e-4, 1e-4, 3e-5, 1e-5, 3e-6, 1e-6]
    for th in thetas:
        s, t = near_degenerate_pair(th, n=1000 if args.quick else 3000)
        a = registration_margins(s, t, nthreads=nth)
        b = registration_margins(s, t, order="reversed", nthreads=nth)
        k = min(len(a["history"]), len(b["history"]))
        out["kappa_sweep"].append({"theta": th, "kappa": a["kappa"], "iterations": [a["num_iterations"], b["num_iterations"]],
                                   "margin": a["margin"], "spread": a["spread"],
                                   "history_max_rel_diff": float(np.max(np.abs(np.array(a["history"][:k]) - np.array(b["history"][:k])) /
                                                                        np.maximum(np.array(a["history"][:k]), 1e-300)))})
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
