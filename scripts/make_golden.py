#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the CPU oracle (run in the authoring container).

The reference ships no fixtures and cannot be built here, so these
vectors pin the ORACLE (regressions) and give the GPU parity tests their expected values;
they do not claim provenance from the reference.  Inputs are regenerated from seeds by
lidar_slam_from_scratch_amd.synth; each fixture stores a checksum of its inputs so that a
drift of the generator is detected rather than silently compared.
"""
import os
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from lidar_slam_from_scratch_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def crc(a):
    return np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()))


def stepwise(src, tgt, max_it, tol, min_err):
    """Python re-enactment of icp.hpp:157-258 from oracle primitives, recording the
    correspondence checksum of every NN pass."""
    tree = orc.KDTree(tgt)
    normals = orc.estimate_normals(tgt, tree, 20)
    cur = src.copy()
    total = np.eye(4)
    prev = np.finfo(np.float64).max
    crcs, hist, conv = [], [], False
    for _ in range(max_it):
        idx, _d = tree.nearest_batch(cur)
        crcs.append(crc(idx))
        sums = orc.normal_equations(cur, tgt[idx], normals[idx])
        err = float(np.sqrt(sums[27] / cur.shape[0]))
        hist.append(err)
        if err < min_err or abs(prev - err) < tol:
            conv = True
            break
        delta = orc.solve_from_sums(sums)
        cur = synth.apply_transform(delta, cur)
        total = delta @ total
        prev = err
    idx, _d = tree.nearest_batch(cur)
    crcs.append(crc(idx))
    return np.array(crcs, dtype=np.uint32), normals


def fixture(name, src, tgt, max_it=50, tol=1e-6, min_err=1e-9, store_inputs=False, steps=True):
    r = orc.icp_point_to_plane(src, tgt, max_it, tol, min_err)
    d = dict(src_crc=crc(src), tgt_crc=crc(tgt), n_src=src.shape[0], n_tgt=tgt.shape[0],
             max_iterations=max_it, tolerance=tol, min_error=min_err,
             transformation=r.transformation, converged=r.converged,
             num_iterations=r.num_iterations, final_error=r.final_error,
             error_history=r.error_history)
    if steps:
        crcs, normals = stepwise(src, tgt, max_it, tol, min_err)
        d["nn_crc"] = crcs
        d["normals_crc"] = crc(normals)
        d["normals_head"] = normals[:32]
    if store_inputs:
        d["source"] = src
        d["target"] = tgt
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print(name, "iters", r.num_iterations, "conv", r.converged, "err", r.final_error)


def widening_fixture():
    """Rows N1/N2 of SURVEY 8(f): voxel filter (file_utils.cpp:148-196) and Scan Context
    (scan_context.hpp:44-142) of a raw synthetic scan, from the oracle."""
    raw = synth.lidar_frame(0, voxel=0, beams=32, azimuths=900)
    vox = orc.voxel_downsample(raw, 0.5)
    other = orc.voxel_downsample(synth.lidar_frame(3, voxel=0, beams=32, azimuths=900), 0.5)
    d = dict(raw_crc=crc(raw), voxel_size=0.5, voxel_rows=vox.shape[0], voxel_crc=crc(vox), voxel_head=vox[:16],
             sc_desc=orc.scan_context(vox), sc_desc_other=orc.scan_context(other),
             sc_distance=orc.scan_context_distance(orc.scan_context(vox), orc.scan_context(other)))
    np.savez_compressed(os.path.join(OUT, "widening.npz"), **d)
    print("widening: voxels", vox.shape[0], "sc distance", d["sc_distance"])


def main():
    os.makedirs(OUT, exist_ok=True)
    widening_fixture()
    s, t, _ = synth.c1_room_corner()
    fixture("c1_room_corner", s, t)
    s, t, _ = synth.kat1_exact_pair()
    fixture("kat1_exact", s, t, store_inputs=True)
    s, t, _ = synth.c2_lidar_pair()
    fixture("c2_lidar_pair", s, t)
    s, t, _ = synth.c3_uniform(20000, seed=14, perm_seed=15)
    fixture("c3_small_20k", s, t, max_it=10, tol=0.0, min_err=0.0)
    if "--full" in sys.argv:
        s, t, _ = synth.c3_uniform()
        fixture("c3_uniform_100k", s, t, max_it=30, tol=0.0, min_err=0.0, steps=False)


if __name__ == "__main__":
    main()
