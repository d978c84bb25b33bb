"""Source-sharded multi-GPU registration: one process per GPU.

The source cloud is cut into contiguous shards, the target (and its normals) is
replicated, and the only per-iteration exchange is the all-reduce of 30 doubles
(21 J^T J + 6 J^T b + sum b^2 + count + the number of ranks whose loop has ended).  Every rank then solves the same 6x6 system from
the same bits, so the pose and the convergence decision need no broadcast.

Transport: RCCL inside the C library (`init_rccl`), bootstrapped by broadcasting the
128-byte unique id over torch.distributed; or host callbacks over any torch.distributed
backend (`init_callbacks`, used with gloo to rehearse the N > 1 path).
"""
import numpy as np


def shard_bounds(n, n_ranks, rank):
    """Contiguous shard [lo, hi) of n rows for `rank`."""
    lo = (n * rank) // n_ranks
    hi = (n * (rank + 1)) // n_ranks
    return lo, hi


def init_rccl(ctx, dist, device=None, allow_single=False):
    """Create the library's RCCL communicator over the ranks of torch.distributed.  A world of
    one needs no exchange and gets none, unless `allow_single` asks for the sharded code path
    anyway (rehearsal of the multi-GPU run on one GPU)."""
    import torch
    n_ranks, rank = dist.get_world_size(), dist.get_rank()
    if n_ranks == 1 and not allow_single:
        return
    ident = [ctx.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(ident, src=0, device=device)
    ctx.comm_init(n_ranks, rank, ident[0])


def init_callbacks(ctx, dist, device=None):
    """Exchange through torch.distributed collectives: on host tensors (gloo), or, with `device`,
    staged through tensors on that device (a process group whose backend only moves device
    memory, i.e. torch's own RCCL: the fallback of bench.py when the library's communicator
    cannot be formed -- slower per exchange, same sums)."""
    import torch
    n_ranks, rank = dist.get_world_size(), dist.get_rank()

    def allreduce(buf):
        t = torch.from_numpy(buf)
        if device is None:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        else:
            d = t.to(device)
            dist.all_reduce(d, op=dist.ReduceOp.SUM)
            t.copy_(d.cpu())

    def allgather(buf, per):
        mine = torch.from_numpy(buf[rank * per:(rank + 1) * per].copy())
        if device is None:
            parts = [torch.empty(per, dtype=torch.float64) for _ in range(n_ranks)]
            dist.all_gather(parts, mine)
        else:
            parts = [torch.empty(per, dtype=torch.float64, device=device) for _ in range(n_ranks)]
            dist.all_gather(parts, mine.to(device))
            parts = [p.cpu() for p in parts]
        for r, p in enumerate(parts):
            buf[r * per:(r + 1) * per] = p.numpy()

    ctx.comm_init_callbacks(n_ranks, rank, allreduce, allgather)


def reduce_normal_equations(local_sums, dist):
    """Host-side statement of the exchange (used by the gloo tests): element-wise sum of
    the 30-vector over ranks; identical bits on every rank afterwards."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(local_sums, dtype=np.float64).copy())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.numpy()


class LocalGroup:
    """N ranks as N threads of ONE process (one context each), the two exchanges done through
    shared host arrays and a barrier: the rehearsal of an N-rank job where N processes cannot
    share the card (the GPU boxes admit six processes on a device).  The sum runs over the
    ranks in rank order on every rank, so every rank reads the same bits -- what the sharded
    path requires of any transport.  A rank that fails breaks the barrier for all."""

    def __init__(self, n_ranks, timeout_s=600.0):
        import threading
        self.n = int(n_ranks)
        self.timeout = timeout_s
        self._barrier = threading.Barrier(self.n)
        self._slots = [None] * self.n
        self.gathered = None          # the last all-gather's result (rank 0's copy): tests look at it
        self.allreduces = 0

    def attach(self, ctx, rank):
        def allreduce(buf):
            self._slots[rank] = buf.copy()
            self._barrier.wait(self.timeout)
            total = self._slots[0].copy()
            for r in range(1, self.n):
                total += self._slots[r]
            self._barrier.wait(self.timeout)   # everybody has read the slots
            buf[:] = total
            if rank == 0:
                self.allreduces += 1

        def allgather(buf, per):
            self._slots[rank] = buf[rank * per:(rank + 1) * per].copy()
            self._barrier.wait(self.timeout)
            for r in range(self.n):
                buf[r * per:(r + 1) * per] = self._slots[r]
            if rank == 0:
                self.gathered = buf.copy()
            self._barrier.wait(self.timeout)

        ctx.comm_init_callbacks(self.n, rank, allreduce, allgather)

    def run(self, fn):
        """fn(rank) on one thread per rank -> list of results; the first exception is re-raised."""
        import threading
        out, errs = [None] * self.n, []

        def body(r):
            try:
                out[r] = fn(r)
            except BaseException as e:  # noqa: BLE001
                errs.append((r, e))
                self._barrier.abort()

        th = [threading.Thread(target=body, args=(r,)) for r in range(self.n)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errs:
            raise errs[0][1]
        return out
