"""Sharding of a job's virtual workers over the GPUs of one node.

The unit of independence of the signal path is the virtual worker (the reference's `tid`: six
scalar Lehmer streams plus one per k-mer, src/sim.c:238-257).  Workers never exchange data, so a
job with T workers on G GPUs gives rank g the contiguous block [g*T/G, (g+1)*T/G) and every batch is
split the same way: no data-path collective.  The only exchange is the start-up broadcast of the
pore-model table from rank 0 (RCCL over xGMI on GPUs; gloo in the CPU tests).
"""
from __future__ import annotations

import numpy as np


def worker_range(rank: int, world: int, num_workers: int):
    """[lo, hi) of the workers owned by `rank`; blocks differ by at most one worker."""
    base, rem = divmod(num_workers, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def owner_of(worker: int, world: int, num_workers: int) -> int:
    base, rem = divmod(num_workers, world)
    cut = rem * (base + 1)
    return worker // (base + 1) if worker < cut else rem + (worker - cut) // max(base, 1)


def batch_workers(n_rec: int, num_workers: int) -> np.ndarray:
    """Worker of each read of a batch under the reference's static partition (src/thread.c:80-99)."""
    if num_workers <= 1:
        return np.zeros(n_rec, np.int32)
    step = (n_rec + num_workers - 1) // num_workers
    return (np.arange(n_rec, dtype=np.int64) // step).astype(np.int32)


def shard_batch(n_rec: int, num_workers: int, rank: int, world: int):
    """(read indices, their global worker ids) of the part of a batch that `rank` processes."""
    wk = batch_workers(n_rec, num_workers)
    lo, hi = worker_range(rank, world, num_workers)
    idx = np.nonzero((wk >= lo) & (wk < hi))[0]
    return idx, wk[idx]


def broadcast_model(mean, stdv, src: int = 0):
    """Broadcast the pore-model table from `src` with torch.distributed (nccl = RCCL, or gloo)."""
    import torch
    import torch.distributed as dist
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.from_numpy(np.stack([np.asarray(mean, np.float32), np.asarray(stdv, np.float32)], 1)).to(dev)
    dist.broadcast(t, src=src)
    h = t.cpu().numpy()
    return np.ascontiguousarray(h[:, 0]), np.ascontiguousarray(h[:, 1])
