"""Several GPUs from ONE host process (INTEGRATION.md "Several GPUs"): one context per GPU over a disjoint block of the job's
virtual workers (sqg_cfg_t.worker_lo / worker_hi), each driven by its own host thread -- the C-host form of what bench.py does
with one process per GPU.  On a one-GPU box the contexts share the device; what must hold is what holds on N: the shards'
signals are, read for read, those of the single context that owns every worker (the reference's src/thread.c:73-116
partition of each batch over the workers)."""
import threading

import numpy as np
import pytest

import orc
from squigulator_amd import api, model, profiles, shard


def _reads(rng, n, lo=300, hi=2500):
    return [bytes(rng.choice(list(b"ACGT"), int(m)).astype(np.uint8)) for m in rng.integers(lo, hi, n)]


@pytest.mark.gpu
@pytest.mark.parametrize("name,T,G,K", [("dna-r9-prom", 4, 2, 400), ("dna-r10-prom", 2, 2, 300), ("dna-r10-prom", 6, 3, 96), ("rna004-prom", 2, 2, 200)])
def test_contexts_on_threads_equal_one_context(name, T, G, K, extra_flags=0):
    rng = np.random.default_rng(31)
    prof, fl = profiles.get_profile(name)
    fl |= extra_flags
    k = profiles.default_kmer_size(fl)
    mean, stdv = model.synthetic_model(k)
    batches = [_reads(rng, K) for _ in range(3)]
    # the whole job in one context
    one = api.SignalGenerator(prof, fl, k, mean, stdv, 42, num_workers=T, mode=api.MODE_CERTIFIED)
    want = []
    for bt in batches:
        b = one.stage(bt, shard.batch_workers(len(bt), T)).run().wait()
        sig = b.signal()
        want.append([sig[b.sig_off[i]:b.sig_off[i + 1]].copy() for i in range(len(bt))])
        b.free()
    one.close()
    # ... and as G contexts, one host thread each, batches queued back to back
    got = [[None] * len(bt) for bt in batches]
    errors = []

    def rank_thread(g):
        try:
            lo, hi = shard.worker_range(g, G, T)
            gen = api.SignalGenerator(prof, fl, k, mean, stdv, 42, num_workers=T, mode=api.MODE_CERTIFIED, worker_lo=lo, worker_hi=hi)
            staged = []
            for bt in batches:
                idx, wk = shard.shard_batch(len(bt), T, g, G)
                staged.append((idx, gen.stage([bt[i] for i in idx], wk)))
            for _, b in staged[:2]:
                b.run()
            for bi, (idx, b) in enumerate(staged):
                b.wait()
                sig = b.signal()
                for j, i in enumerate(idx):
                    got[bi][i] = sig[b.sig_off[j]:b.sig_off[j + 1]].copy()
                b.free()
                if bi + 2 < len(staged):
                    staged[bi + 2][1].run()
            gen.close()
        except Exception as e:  # noqa: BLE001
            errors.append((g, repr(e)))

    th = [threading.Thread(target=rank_thread, args=(g,)) for g in range(G)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    for bi in range(len(batches)):
        for i in range(len(batches[bi])):
            np.testing.assert_array_equal(got[bi][i], want[bi][i], err_msg=f"batch {bi} read {i}")


@pytest.mark.gpu
@pytest.mark.parametrize("rep", range(3))
def test_contexts_on_threads_with_per_link_rows(rep, monkeypatch):
    """the same with the per-link rows of the 5^8 / 5^9 tables and of many workers with few reads (forced here: SQG_NO_PART, development
    library): k_link_prefix's groups read the worker's row that its first group rewrites -- with a second context's kernels on the same
    CUs a group that read it late saw the advanced row (5 failures in 8 runs before the read moved in front of the barrier)"""
    monkeypatch.setenv("SQG_NO_PART", "1")
    test_contexts_on_threads_equal_one_context("dna-r10-prom", 2, 2, 300)


# the fall-back paths under the same load -- two contexts' kernels sharing the CUs is what found the per-link rows' race -- (see
# tests/test_fuzz_parity.py VARIANTS)
@pytest.mark.gpu
@pytest.mark.parametrize("name,T,G,K", [("dna-r10-prom", 2, 2, 300), ("dna-r9-prom", 4, 2, 400)])
@pytest.mark.parametrize("variant", ["order-free", "per-link-rows", "no-precount", "wg-per-link"])
def test_contexts_on_threads_on_the_fallback_paths(variant, name, T, G, K, monkeypatch):
    env = {"per-link-rows": {"SQG_NO_PART": "1"}, "no-precount": {"SQG_NO_PRECOUNT": "1"}, "wg-per-link": {"SQG_PART_WG_EVENTS": "1"}}.get(variant, {})
    for kname, val in env.items():
        monkeypatch.setenv(kname, val)
    test_contexts_on_threads_equal_one_context(name, T, G, K, extra_flags=profiles.SQ_ORDER_FREE if variant == "order-free" else 0)


@pytest.mark.gpu
def test_sharded_contexts_match_the_oracle():
    """the same against the oracle's single-process run (-t 4), two contexts, sequentially driven"""
    rng = np.random.default_rng(5)
    prof, fl = profiles.get_profile("dna-r9-prom")
    mean, stdv = model.synthetic_model(6)
    T, G = 4, 2
    batches = [_reads(rng, 64, 200, 900) for _ in range(2)]
    orac = orc.Oracle(prof, fl, 6, mean, stdv, 42, num_workers=T)
    want = [orac.run_batch_seqs(bt, want_ss=False) for bt in batches]
    orac.close()
    gens = [api.SignalGenerator(prof, fl, 6, mean, stdv, 42, num_workers=T, mode=api.MODE_EXACT,
                                worker_lo=shard.worker_range(g, G, T)[0], worker_hi=shard.worker_range(g, G, T)[1]) for g in range(G)]
    for bi, bt in enumerate(batches):
        for g, gen in enumerate(gens):
            idx, wk = shard.shard_batch(len(bt), T, g, G)
            b = gen.stage([bt[i] for i in idx], wk).run().wait()
            sig = b.signal()
            for j, i in enumerate(idx):
                np.testing.assert_array_equal(sig[b.sig_off[j]:b.sig_off[j + 1]], want[bi][i].sig)
                assert b.offset[j] == want[bi][i].offset
            b.free()
    for gen in gens:
        gen.close()
