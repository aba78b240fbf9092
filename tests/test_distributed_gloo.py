"""N>1 path on CPU: two -- and eight, the node's size -- processes over gloo.  Rank 0 owns the pore model and broadcasts it; every
rank processes only the reads of the workers it owns (squigulator_amd.shard); gathering the shards
must reproduce the single-process run bit for bit -- i.e. sharding workers over GPUs changes nothing.
The per-rank compute stands in for the GPU kernel by calling the oracle (this is a test)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _reads(n, seed):
    rng = np.random.default_rng(seed)
    return [bytes(rng.choice(list(b"ACGT"), size=int(L)).astype(np.uint8)) for L in rng.integers(250, 900, size=n)]


def _worker(rank, world, port, T, n_batches, K, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import orc
    from squigulator_amd import model, profiles, shard
    prof, fl = profiles.get_profile("dna-r9-prom")
    if rank == 0:
        mean, stdv = model.synthetic_model(6)
    else:
        mean, stdv = np.zeros(4096, np.float32), np.zeros(4096, np.float32)
    mean, stdv = shard.broadcast_model(mean, stdv, src=0)
    o = orc.Oracle(prof, fl, 6, mean, stdv, 42, num_workers=T)
    digest = []
    for b in range(n_batches):
        reads = _reads(K, 100 + b)
        idx, wk = shard.shard_batch(K, T, rank, world)
        res = o.run_batch_assigned([reads[i] for i in idx], wk, want_ss=False)
        for i, r in zip(idx, res):
            digest.append((b, int(i), len(r.sig), int(np.int64(r.sig.astype(np.int64) @ np.arange(1, len(r.sig) + 1) % 1000003)), r.offset))
    gathered = [None] * world
    dist.all_gather_object(gathered, digest)
    if rank == 0:
        out.put(sorted(sum(gathered, [])))
    dist.barrier()
    dist.destroy_process_group()
    o.close()


@pytest.mark.parametrize("world,T,K", [(2, 8, 8), (2, 6, 13), (8, 8, 8), (8, 8, 21), (8, 11, 40)])
def test_sharding_by_worker_equals_single_process(world, T, K):
    sys.path.insert(0, HERE)
    import orc
    from squigulator_amd import model, profiles, shard
    n_batches = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, T, n_batches, K, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    prof, fl = profiles.get_profile("dna-r9-prom")
    mean, stdv = model.synthetic_model(6)
    o = orc.Oracle(prof, fl, 6, mean, stdv, 42, num_workers=T)
    want = []
    for b in range(n_batches):
        reads = _reads(K, 100 + b)
        res = o.run_batch_assigned(reads, shard.batch_workers(K, T), want_ss=False)
        for i, r in enumerate(res):
            want.append((b, i, len(r.sig), int(np.int64(r.sig.astype(np.int64) @ np.arange(1, len(r.sig) + 1) % 1000003)), r.offset))
    o.close()
    assert got == sorted(want)


def _stream_counts(reads, results, k, T, worker):
    """samples each (worker, k-mer) stream is asked for by these reads: what sqg_batch_run_begin leaves on the device"""
    code = np.zeros(256, np.int64)
    code[[ord(x) for x in "Cc"]], code[[ord(x) for x in "Gg"]], code[[ord(x) for x in "Tt"]] = 1, 2, 3
    c = np.zeros((T, 4 ** k), np.uint64)
    for r, res, w in zip(reads, results, worker):
        b = code[np.frombuffer(r, np.uint8)]
        ranks = np.zeros(len(b) - k + 1, np.int64)
        for i in range(k):                                        # src/seq.h:31-42: the first base is the most significant
            ranks = ranks * 4 + b[i:len(b) - k + 1 + i]
        np.add.at(c[w], ranks, np.asarray(res.ss, np.uint64))
    return c.reshape(-1)


def _range_worker(rank, world, port, T, K, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import torch
    import orc
    from squigulator_amd import model, profiles, shard
    prof, fl = profiles.get_profile("dna-r9-prom")
    mean, stdv = model.synthetic_model(6)
    o = orc.Oracle(prof, fl, 6, mean, stdv, 42, num_workers=T)
    reads = _reads(K, 7)
    wk = shard.batch_workers(K, T)
    res = o.run_batch_assigned(reads, wk, want_ss=True)       # every rank knows the whole batch here; it owns [lo, hi)
    lo, hi = shard.read_range(rank, world, K)
    mine = _stream_counts(reads[lo:hi], res[lo:hi], 6, T, wk[lo:hi])
    t = torch.from_numpy(mine.astype(np.uint32).view(np.int32))
    before, after = shard.exchange_counts(t)
    out.put((rank, lo, hi, before.numpy().view(np.uint32).copy(), after.numpy().view(np.uint32).copy()))
    dist.barrier()
    dist.destroy_process_group()
    o.close()


@pytest.mark.parametrize("world,T,K", [(2, 1, 9), (2, 2, 7), (8, 1, 19), (8, 3, 8)])
def test_range_sharding_exchange_gives_every_rank_the_counts_of_the_other_ranges(world, T, K):
    """world sizes 2 and 8 over gloo (8: the all-gather of eight count vectors, a rank whose range is a single read): `before` is exactly what the reads ahead of a rank's range draw from each stream,
    `after` what the reads behind it draw"""
    sys.path.insert(0, HERE)
    import orc
    from squigulator_amd import model, profiles, shard
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_range_worker, args=(r, world, port, T, K, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    prof, fl = profiles.get_profile("dna-r9-prom")
    mean, stdv = model.synthetic_model(6)
    o = orc.Oracle(prof, fl, 6, mean, stdv, 42, num_workers=T)
    reads = _reads(K, 7)
    wk = shard.batch_workers(K, T)
    res = o.run_batch_assigned(reads, wk, want_ss=True)
    o.close()
    ranges = sorted((lo, hi) for _, lo, hi, _, _ in got)
    assert ranges[0][0] == 0 and ranges[-1][1] == K and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    for rank, lo, hi, before, after in got:
        np.testing.assert_array_equal(before, _stream_counts(reads[:lo], res[:lo], 6, T, wk[:lo]).astype(np.uint32))
        np.testing.assert_array_equal(after, _stream_counts(reads[hi:], res[hi:], 6, T, wk[hi:]).astype(np.uint32))
