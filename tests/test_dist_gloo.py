"""The N > 1 path on CPU: world_size-2 gloo.  Checks the host-side sharding logic and the
one exchange of the path (all-reduce of the 30-double normal-equation vector: 28 sums, the count and the
number of ranks whose loop has ended): sharded
sums -> identical bits on every rank -> same pose as the unsharded run.  The per-shard
arithmetic is done by the oracle here (no GPU in this container); the same sharded path
through the HIP library is exercised by tests/test_gpu_dist.py on the GPU box."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from lidar_slam_from_scratch_amd import dist as icpdist, synth
    from oracle import oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    src, tgt, _ = synth.c1_room_corner(3001)  # odd size: ragged shards
    lo, hi = icpdist.shard_bounds(src.shape[0], world, rank)
    tree = orc.KDTree(tgt)
    normals = orc.estimate_normals(tgt, tree, 20)
    cur = src[lo:hi].copy()
    total = np.eye(4)
    prev = np.finfo(np.float64).max
    hist, conv = [], False
    for _ in range(50):  # icp.hpp:181-232 with the sums exchanged
        idx, _d = tree.nearest_batch(cur)
        local = np.zeros(30)  # [29]: ranks whose loop has ended (none while this loop runs)
        local[:28] = orc.normal_equations(cur, tgt[idx], normals[idx])
        local[28] = cur.shape[0]
        sums = icpdist.reduce_normal_equations(local, dist)
        err = float(np.sqrt(sums[27] / sums[28]))
        hist.append(err)
        if err < 1e-9 or abs(prev - err) < 1e-6:
            conv = True
            break
        delta = orc.solve_from_sums(sums[:28])
        cur = synth.apply_transform(delta, cur)
        total = delta @ total
        prev = err
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), T=total, hist=np.array(hist), conv=conv,
             lo=lo, hi=hi, count=sums[28])
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_cover_and_are_contiguous():
    from lidar_slam_from_scratch_amd import dist as icpdist
    for n in (1, 7, 100000, 100003):
        for w in (1, 2, 3, 8):
            b = [icpdist.shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def test_sharded_icp_world2_gloo(tmp_path, oracle):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % k)) for k in range(world)]
    assert (r[0]["T"] == r[1]["T"]).all()          # identical bits on every rank, no broadcast
    assert (r[0]["hist"] == r[1]["hist"]).all()
    assert r[0]["count"] == 3001 and r[0]["hi"] == r[1]["lo"]
    from lidar_slam_from_scratch_amd import synth
    src, tgt, _ = synth.c1_room_corner(3001)
    ref = oracle.icp_point_to_plane(src, tgt)
    assert bool(r[0]["conv"]) == ref.converged and len(r[0]["hist"]) == ref.loop_iterations
    dt, dr = synth.pose_delta(r[0]["T"], ref.transformation)
    assert dt < 1e-10 and dr < 1e-10               # only the summation order differs
    np.testing.assert_allclose(r[0]["hist"], ref.error_history[:-1], atol=1e-12)


def test_local_group_exchanges_are_rank_ordered_and_identical():
    """dist.LocalGroup (N ranks as N threads of one process: what the C4 rehearsal on one GPU runs on): the all-reduce
    adds the ranks' vectors in rank order on every rank (identical bits everywhere, the property the sharded path
    needs of any transport), the all-gather leaves every rank's slice at its rank's offset.  Driven here by a stand-in
    for the context: an object whose comm_init_callbacks keeps the two callbacks."""
    from lidar_slam_from_scratch_amd import dist as icpdist

    class FakeCtx:
        def comm_init_callbacks(self, n_ranks, rank, allreduce, allgather):
            self.n_ranks, self.rank, self.allreduce, self.allgather = n_ranks, rank, allreduce, allgather

    world = 8
    group = icpdist.LocalGroup(world, timeout_s=30.0)
    rng = np.random.default_rng(3)
    vecs = rng.normal(0, 1, (world, 30)) * 10.0 ** rng.integers(-8, 9, (world, 1))     # sums whose order matters

    def body(rank):
        ctx = FakeCtx()
        group.attach(ctx, rank)
        out = []
        for rep in range(3):
            buf = vecs[rank].copy() * (rep + 1)
            ctx.allreduce(buf)
            out.append(buf)
        gat = np.zeros(world * 5)
        gat[rank * 5:(rank + 1) * 5] = rank + 0.25
        ctx.allgather(gat, 5)
        return out, gat

    res = group.run(body)
    for rep in range(3):
        want = vecs[0].copy() * (rep + 1)
        for r in range(1, world):
            want = want + vecs[r] * (rep + 1)                         # rank order, left to right
        for rank in range(world):
            assert (res[rank][0][rep] == want).all()
    for rank in range(world):
        assert (res[rank][1] == np.repeat(np.arange(world) + 0.25, 5)).all()
    assert group.allreduces == 3

    def failing(rank):                                               # a rank that dies must not leave the others waiting
        ctx = FakeCtx()
        group2.attach(ctx, rank)
        if rank == 3:
            raise RuntimeError("rank 3 gives up")
        ctx.allreduce(np.zeros(4))

    group2 = icpdist.LocalGroup(world, timeout_s=30.0)
    with pytest.raises(Exception):
        group2.run(failing)
