"""The first event pass of batch i+1 inside the launch sequence of batch i (k_part_hand_count: next to batch i's stream hand-out), taken
whenever batch i+1 is staged by the time batch i is run.  Same int16, same dwells, same validity of results as the plain sequence;
a batch that was counted ahead and is then freed without a run leaves nothing behind.  src/gensig.c:249-272."""
import numpy as np
import pytest

import orc
from squigulator_amd import api, model, profiles


def _reads(rng, n, lo, hi):
    return [bytes(rng.choice(list(b"ACGT"), int(m)).astype(np.uint8)) for m in rng.integers(lo, hi, n)]


def _check(b, want, tag):
    sig, dw = b.signal(), b.dwell()
    for i, w in enumerate(want):
        np.testing.assert_array_equal(sig[b.sig_off[i]:b.sig_off[i + 1]], w.sig, err_msg=f"{tag} read {i}")
        np.testing.assert_array_equal(dw[b.ev_off[i]:b.ev_off[i + 1]], w.ss, err_msg=f"{tag} read {i} (dwell)")
        assert b.offset[i] == w.offset and b.median_before[i] == w.median_before


@pytest.mark.gpu
@pytest.mark.parametrize("name,T,lens", [("dna-r10-prom", 1, (300, 4000)), ("dna-r10-prom", 3, (200, 2500)), ("rna004-prom", 1, (300, 2500)),
                                         ("dna-r10-prom", 1, (9000, 30000)),       # reads cut into pieces of several links (totals added up)
                                         ("dna-r9-prom", 1, (300, 4000)), ("dna-r9-prom", 3, (9000, 30000)), ("rna-r9-prom", 2, (300, 2500))],   # k <= 6: one partition, ONE event pass
                         ids=["r10_t1", "r10_t3", "rna004_t1", "r10_long_reads", "r9_t1", "r9_t3_long_reads", "rna9_t2"])
@pytest.mark.parametrize("mode", [api.MODE_CERTIFIED, api.MODE_EXACT], ids=["certified", "exact"])
def test_batches_staged_ahead_equal_the_oracle(name, T, lens, mode):
    rng = np.random.default_rng(99)
    prof, fl = profiles.get_profile(name)
    k = profiles.default_kmer_size(fl)
    mean, stdv = model.synthetic_model(k)
    nread = 40 if lens[1] > 20000 else 260
    batches = [_reads(rng, nread + 17 * i, *lens) for i in range(5)]
    orac = orc.Oracle(prof, fl, k, mean, stdv, 7, num_workers=T)
    want = [orac.run_batch_seqs(bt, want_ss=True) for bt in batches]
    orac.close()
    gen = api.SignalGenerator(prof, fl, k, mean, stdv, 7, num_workers=T, mode=mode)
    staged = [gen.stage(bt) for bt in batches]                      # everything staged before the first run: every run finds its successor
    for b in staged:
        b.run()
    for i, b in enumerate(staged):
        b.wait()
        if i >= len(staged) - 2:                                    # (results stay valid until two more batches have been run)
            _check(b, want[i], f"batch {i}")
        else:
            with pytest.raises(api.SqgError):
                b.signal()
    for b in staged:
        b.free()
    # the same job, each batch consumed while the next two are staged / queued (the streaming pattern of bench.py)
    gen.close()
    gen = api.SignalGenerator(prof, fl, k, mean, stdv, 7, num_workers=T, mode=mode)
    cur = gen.stage(batches[0]).run()
    nxt = gen.stage(batches[1])
    for i in range(len(batches)):
        nn = gen.stage(batches[i + 2]) if i + 2 < len(batches) else None
        if nxt is not None:
            nxt.run()
        cur.wait()
        _check(cur, want[i], f"streaming, batch {i}")
        cur.free()
        cur, nxt = nxt, nn
    gen.close()


@pytest.mark.gpu
def test_precount_is_taken_and_can_be_switched_off(monkeypatch, capfd):
    """the development build's SQG_NO_PRECOUNT: the plain sequence; both give the same bytes.  That the fused launch is taken is read off the
    library's own SQG_VERBOSE lines (one per batch, the first eight batches of a context)"""
    rng = np.random.default_rng(5)
    prof, fl = profiles.get_profile("dna-r10-prom")
    k = profiles.default_kmer_size(fl)
    mean, stdv = model.synthetic_model(k)
    batches = [_reads(rng, 600, 2000, 9000) for _ in range(4)]
    monkeypatch.setenv("SQG_VERBOSE", "1")
    out, ahead = {}, {}
    for off in (False, True):
        if off:
            monkeypatch.setenv("SQG_NO_PRECOUNT", "1")
        capfd.readouterr()
        gen = api.SignalGenerator(prof, fl, k, mean, stdv, 3, num_workers=1, mode=api.MODE_CERTIFIED)
        staged = [gen.stage(bt) for bt in batches]
        res = []
        for i in range(0, 4, 2):
            staged[i].run(); staged[i + 1].run()
            for b in staged[i:i + 2]:
                b.wait()
                res.append((b.signal().copy(), b.dwell().copy()))
        for b in staged:
            b.free()
        gen.close()
        out[off] = res
        ahead[off] = capfd.readouterr().err.count("first event pass: ran ahead")
    for (s0, d0), (s1, d1) in zip(out[False], out[True]):
        np.testing.assert_array_equal(s0, s1)
        np.testing.assert_array_equal(d0, d1)
    assert ahead == {False: 3, True: 0}, ahead                      # every batch but the first had its pass run ahead


@pytest.mark.gpu
def test_a_batch_counted_ahead_and_then_freed_leaves_nothing_behind():
    """stage b0, b1, b2; run b0 (b1's first pass rides along); free b1 without running it; b2 must come out as it does when b1 is
    abandoned in the plain sequence (a context that never saw b1's successor staged ahead)"""
    rng = np.random.default_rng(17)
    prof, fl = profiles.get_profile("dna-r10-prom")
    k = profiles.default_kmer_size(fl)
    mean, stdv = model.synthetic_model(k)
    b0, b1, b2, b3 = (_reads(rng, 220, 9000, 26000) for _ in range(4))     # (split reads: the totals are added up with atomics)
    sigs = []
    for ahead in (True, False):
        gen = api.SignalGenerator(prof, fl, k, mean, stdv, 21, num_workers=1, mode=api.MODE_CERTIFIED)
        s0 = gen.stage(b0)
        s1 = gen.stage(b1)
        if ahead:
            s0.run()                                                # s1 is staged: counted ahead
            s1.free()
        else:
            s1.free()                                               # nothing staged behind s0 when it runs
            s0.run()
        s2 = gen.stage(b2)
        s3 = gen.stage(b3)
        s2.run(); s3.run()
        got = []
        for s in (s0, s2, s3):
            s.wait()
        for s in (s2, s3):
            got.append((s.signal().copy(), s.dwell().copy(), np.array(s.sig_off)))
        sigs.append(got)
        for s in (s0, s2, s3):
            s.free()
        gen.close()
    for a, b in zip(*sigs):
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)


@pytest.mark.gpu
@pytest.mark.parametrize("T", [1, 3])
def test_draws_made_ahead_are_the_draws(T):
    """few workers: a thread of the context makes the per-read `offset` / `median_before` draws (host libm) ahead of staging; staging
    takes what is ready and draws the rest -- whatever the split, the doubles are the oracle's.  Small batches with pauses (the ring
    runs ahead), one batch larger than the ring (prefix from the ring, the rest shared by the helper threads), sqg_skip_reads in
    between (the streams move another way: the ring starts over)."""
    import time
    rng = np.random.default_rng(123)
    prof, fl = profiles.get_profile("dna-r9-prom")
    k = profiles.default_kmer_size(fl)
    mean, stdv = model.synthetic_model(k)
    sizes = [300, 17, 1200, 300, 20000, 5, 900, 300]
    batches = [_reads(rng, m, 20, 60) for m in sizes]
    orac = orc.Oracle(prof, fl, k, mean, stdv, 11, num_workers=T)
    want = [orac.run_batch_seqs(bt, want_ss=False) for bt in batches]
    orac.close()
    gen = api.SignalGenerator(prof, fl, k, mean, stdv, 11, num_workers=T, mode=api.MODE_CERTIFIED)
    for bi, bt in enumerate(batches):
        b = gen.submit(bt)
        for i, w in enumerate(want[bi]):
            assert b.offset[i] == w.offset and b.median_before[i] == w.median_before, (bi, i)
        sig = b.signal()
        for i in (0, len(bt) // 2, len(bt) - 1):
            np.testing.assert_array_equal(sig[b.sig_off[i]:b.sig_off[i + 1]], want[bi][i].sig)
        b.free()
        time.sleep(0.03)                                            # the draw-ahead thread fills its ring
    gen.close()
