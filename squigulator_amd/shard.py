"""Sharding of a job's virtual workers over the GPUs of one node.

The unit of independence of the signal path is the virtual worker (the reference's `tid`: six
scalar Lehmer streams plus one per k-mer, src/sim.c:238-257).  Workers never exchange data, so a
job with T workers on G GPUs gives rank g the contiguous block [g*T/G, (g+1)*T/G) and every batch is
split the same way: no data-path collective.  The only exchange is the start-up broadcast of the
pore-model table from rank 0 (RCCL over xGMI on GPUs; gloo in the CPU tests).

Range sharding (include/sqg.h, SURVEY.md section 8e "strict -t 1"): with fewer workers than GPUs every rank owns all
workers and generates the reads [lo, hi) of each batch (`read_range`); the one exchange step of the path is then the
all-gather of the per-(worker, k-mer) sample counts of every rank's range (`exchange_counts`).
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


def read_range(rank: int, world: int, n_rec: int):
    """[lo, hi) of the reads of a batch that `rank` generates under range sharding (contiguous, in batch order)."""
    return n_rec * rank // world, n_rec * (rank + 1) // world


class _DevArray:
    """a raw device pointer dressed for torch.as_tensor (zero-copy)"""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i4", "data": (ptr, False), "version": 2}


def counts_tensor(ptr: int, n: int, device):
    """the counts sqg_batch_run_begin left on the device, as an int32 torch tensor (same bits as the uint32 counts)"""
    import torch
    return torch.as_tensor(_DevArray(ptr, n), device=device)


def exchange_counts(mine):
    """All-gather the per-stream sample counts of every rank's range (rank order = range order) and return
    (before, after): element-wise sums over the earlier and the later ranks, as int32 tensors on `mine`'s device.
    uint32 arithmetic is done on the int32 bit patterns (two's complement wrap-around is the same addition)."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    on_cpu = dist.get_backend() != "nccl"
    t = mine.cpu() if on_cpu else mine
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t.contiguous())
    zero = torch.zeros_like(t, dtype=torch.int64)
    before = sum((p.to(torch.int64) for p in parts[:rank]), zero).to(torch.int32)
    after = sum((p.to(torch.int64) for p in parts[rank + 1:]), zero).to(torch.int32)
    return before.to(mine.device).contiguous(), after.to(mine.device).contiguous()
