"""Few workers, many reads (the reference's `-t 1` regime, which every golden in its test/ directory uses): the worker
chains are cut into links that k_events walks concurrently (k_link_hist, k_link_prefix).  The result has to be the
one the serial walk gives -- the oracle's -- whatever the cut."""
import numpy as np
import pytest

import orc
from squigulator_amd import api, model, profiles

FLAG_SETS = [0, profiles.SQ_PREFIX, profiles.SQ_RNA | profiles.SQ_PREFIX, profiles.SQ_IDEAL_TIME, profiles.SQ_RNA]


def _reads(rng, n, k, longest):
    lens = rng.choice([1, k - 1, k, 64, 65, 200, 513, 1025, longest], n)
    return [bytes(rng.choice(list(b"ACGTN"), int(m), p=[.245, .245, .245, .245, .02]).astype(np.uint8)) for m in lens]


def _check(prof, flags, k, T, s, batches, modes=(api.MODE_CERTIFIED, api.MODE_EXACT), salt=0):
    mean, stdv = model.synthetic_model(k, salt=salt)
    orac = orc.Oracle(prof, flags, k, mean, stdv, s, num_workers=T)
    want = [orac.run_batch_seqs(bt) for bt in batches]
    orac.close()
    for mode in modes:
        gen = api.SignalGenerator(prof, flags, k, mean, stdv, s, num_workers=T, mode=mode)
        for bi, bt in enumerate(batches):
            b = gen.submit(bt)
            sig, dw = b.signal(), b.dwell()
            for i, w in enumerate(want[bi]):
                np.testing.assert_array_equal(sig[b.sig_off[i]:b.sig_off[i + 1]], w.sig, err_msg=f"mode {mode} batch {bi} read {i}")
                np.testing.assert_array_equal(dw[b.ev_off[i]:b.ev_off[i + 1]], w.ss)
                assert b.offset[i] == w.offset and b.median_before[i] == w.median_before
            b.free()
        gen.close()


@pytest.mark.gpu
@pytest.mark.parametrize("links", ["1", "5", "100000"])
@pytest.mark.parametrize("seed", range(10))
def test_any_cut_of_the_worker_chains_gives_the_serial_result(seed, links, monkeypatch):
    monkeypatch.setenv("SQG_SPLIT_CHAINS", links)
    rng = np.random.default_rng(7000 + seed)
    name = ["dna-r9-prom", "dna-r10-prom", "rna004-prom", "rna-r9-prom", "dna-r9-min"][seed % 5]
    prof, fl = profiles.get_profile(name)
    k = profiles.default_kmer_size(fl)
    flags = fl | FLAG_SETS[seed % len(FLAG_SETS)] if not (fl & profiles.SQ_RNA) else fl | (profiles.SQ_PREFIX if seed % 2 else 0)
    T = int(rng.choice([1, 1, 2, 3]))
    batches = [_reads(rng, int(rng.integers(T + 1, 6 * T + 12)), k, 3000) for _ in range(3)]
    _check(prof, flags, k, T, int(rng.integers(1, 1 << 30)), batches, salt=seed)


@pytest.mark.gpu
@pytest.mark.parametrize("name,T", [("dna-r9-prom", 1), ("dna-r10-prom", 1), ("dna-r9-prom", 8), ("rna004-prom", 2)])
def test_t1_regime_is_split_by_default_and_matches_the_oracle(name, T):
    """enough events for the heuristic to cut the chains (no environment override); two batches, carried state"""
    rng = np.random.default_rng(99)
    prof, fl = profiles.get_profile(name)
    k = profiles.default_kmer_size(fl)
    if fl & profiles.SQ_RNA:
        fl |= profiles.SQ_PREFIX
    batches = [_reads(rng, 160, k, 4000) for _ in range(2)]
    assert sum(len(r) for r in batches[0]) > 70000
    _check(prof, fl, k, T, 42, batches, modes=(api.MODE_CERTIFIED,))


@pytest.mark.gpu
def test_split_and_unsplit_runs_agree_at_size(monkeypatch):
    """-t 4 with 4000 reads per batch: the split walk against the serial one, every int16"""
    rng = np.random.default_rng(5)
    prof, fl = profiles.get_profile("dna-r9-prom")
    mean, stdv = model.synthetic_model(6)
    batches = [[bytes(rng.choice(list(b"ACGT"), int(m)).astype(np.uint8)) for m in rng.integers(500, 3000, 4000)] for _ in range(2)]
    out = {}
    for setting in ("0", None):
        if setting is None:
            monkeypatch.delenv("SQG_SPLIT_CHAINS", raising=False)
        else:
            monkeypatch.setenv("SQG_SPLIT_CHAINS", setting)
        gen = api.SignalGenerator(prof, fl, 6, mean, stdv, 42, num_workers=4, mode=api.MODE_CERTIFIED)
        res = []
        for bt in batches:
            b = gen.submit(bt)
            res.append((b.signal().copy(), np.array(b.sig_off).copy()))
            b.free()
        gen.close()
        out[setting] = res
    for (s0, o0), (s1, o1) in zip(out["0"], out[None]):
        np.testing.assert_array_equal(o0, o1)
        np.testing.assert_array_equal(s0, s1)


@pytest.mark.gpu
@pytest.mark.parametrize("turns", ["2", "3"])
@pytest.mark.parametrize("T", [1, 5])
def test_nine_mer_sample_counts_wrap_correctly(turns, T, monkeypatch):
    """k > 6 keeps, per stream, the number of samples drawn so far; the state depends on it mod (M-1)/2 only.  The hook
    starts every count at 2 or 3 times that (the same stream position): 2 exercises the top of the jump tables, 3 makes
    the first batch normalise the counts.  Output must not change."""
    monkeypatch.setenv("SQG_TEST_ROW_TURNS", turns)
    rng = np.random.default_rng(31)
    prof, fl = profiles.get_profile("dna-r10-prom")
    batches = [_reads(rng, 3 * T + 4, 9, 2000) for _ in range(3)]
    _check(prof, fl, 9, T, 1234, batches)


@pytest.mark.gpu
@pytest.mark.parametrize("T", [1, 2, 9])
def test_very_ragged_batches(T):
    """reads of 1 base next to reads of 2*10^5: uneven links (T < reads), a single long chain per workgroup (T = reads),
    segment tails of every length"""
    rng = np.random.default_rng(77)
    prof, fl = profiles.get_profile("dna-r9-prom")
    lens = [200000, 5, 64, 150000, 1, 3000, 511, 512, 513]
    batches = [[bytes(rng.choice(list(b"ACGT"), m).astype(np.uint8)) for m in rng.permutation(lens)] for _ in range(2)]
    _check(prof, fl, 6, T, 11, batches, modes=(api.MODE_CERTIFIED,))
    prof, fl = profiles.get_profile("dna-r10-prom")
    _check(prof, fl, 9, T, 11, batches[:1], modes=(api.MODE_CERTIFIED,))


@pytest.mark.gpu
@pytest.mark.parametrize("slice_len", [None, "1024", "5000"])
@pytest.mark.parametrize("name,T", [("dna-r10-prom", 1), ("rna004-prom", 1), ("rna004-prom", 3)])
def test_streams_with_many_events_in_a_row(name, T, slice_len, monkeypatch):
    """homopolymers and a sequence every read carries (poly-A tail + adaptor with --prefix=yes): hundreds of consecutive events
    of a batch fall on ONE k-mer stream -- the bucketed hand-out takes such a stream as a whole (k_part_hand), and a partition
    of the ranks that holds many events is cut into more slices than the others (k_part_slices; any slice length must do)"""
    if slice_len:
        monkeypatch.setenv("SQG_PART_SLICE", slice_len)
    rng = np.random.default_rng(123)
    prof, fl = profiles.get_profile(name)
    if fl & profiles.SQ_RNA:
        fl |= profiles.SQ_PREFIX
    common = bytes(rng.choice(list(b"ACGT"), 90).astype(np.uint8))
    batches = []
    for _ in range(2):
        reads = []
        for i in range(150):
            body = bytes(rng.choice(list(b"ACGT"), int(rng.integers(100, 700))).astype(np.uint8))
            reads.append(b"A" * int(rng.integers(50, 400)) + body + common + b"T" * int(rng.integers(0, 300)) + b"AC" * int(rng.integers(0, 60)))
        batches.append(reads)
    assert sum(len(r) for r in batches[0]) > 70000
    _check(prof, fl, 9, T, 77, batches, modes=(api.MODE_CERTIFIED,))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["dna-r10-prom", "rna004-prom"])
@pytest.mark.parametrize("extra", [0, profiles.SQ_IDEAL_TIME, profiles.SQ_PREFIX, profiles.SQ_PREFIX | profiles.SQ_IDEAL_TIME],
                         ids=["plain", "ideal_time", "prefix", "prefix_ideal_time"])
@pytest.mark.parametrize("T", [1, 2])
@pytest.mark.parametrize("hand", ["ordered", "claims"])
def test_bucketed_hand_out_with_every_option(name, extra, T, hand, monkeypatch):
    """9-mer tables, chains cut by the default heuristic: constant dwell (--ideal-time), DNA/RNA prefixes (two segments per read),
    both arithmetic modes, two batches; the hand-out by ordered LDS atomics (k_part_hand_ord, what a device that passes
    k_lds_order_check runs) and by the claim protocol (k_part_hand, the fall-back)"""
    if hand == "claims":
        monkeypatch.setenv("SQG_PART_CLAIMS", "1")
    rng = np.random.default_rng(2024 + T)
    prof, fl = profiles.get_profile(name)
    batches = [_reads(rng, 170, 9, 3500) for _ in range(2)]
    assert sum(max(len(r) - 8, 5) for r in batches[0]) > 66000
    _check(prof, fl | extra, 9, T, 31, batches)


@pytest.mark.gpu
def test_lds_atomics_are_served_in_lane_order(monkeypatch):
    """the property the few-worker hand-out rests on (Kernels): measured at create, and here at a larger size with eight
    wavefronts per CU competing for the LDS: 2*10^8 fetch-adds, not one out of order.  SQG_PART_CLAIMS=1: the context does not
    rely on it (and the kernels it then uses are what the `claims` cases of this file run)"""
    prof, fl = profiles.get_profile("dna-r10-prom")
    mean, stdv = model.synthetic_model(9)
    gen = api.SignalGenerator(prof, fl, 9, mean, stdv, 1, num_workers=1)
    bad, used = gen.probe_lds_order(workgroups=2048, rounds=96)
    gen.close()
    assert bad == 0 and used
    monkeypatch.setenv("SQG_PART_CLAIMS", "1")
    gen = api.SignalGenerator(prof, fl, 9, mean, stdv, 1, num_workers=1)
    bad, used = gen.probe_lds_order(workgroups=64, rounds=4)
    gen.close()
    assert bad == 0 and not used


@pytest.mark.gpu
@pytest.mark.parametrize("links", [None, "100000"])
@pytest.mark.parametrize("name", ["rna004-prom", "rna-r9-prom"])
def test_second_part_starts_at_every_position_of_a_segment(name, links, monkeypatch):
    """RNA with --prefix=yes: a read is [transcript + poly-A + adaptor] then the stall, whose k-mers start k - 1 bases further on
    (src/genread.c:87-88).  Read lengths chosen so that the first event of the second part falls on, just before and just behind the
    512-event segment boundaries of k_part_events -- which are also where long reads are cut into pieces (forced here: one piece per
    segment) -- and on the 64-event tile boundaries in between"""
    if links:
        monkeypatch.setenv("SQG_SPLIT_CHAINS", links)
    rng = np.random.default_rng(4242)
    prof, fl = profiles.get_profile(name)
    fl |= profiles.SQ_PREFIX
    k = profiles.default_kmer_size(fl)
    tail = 158 + 79 - (k - 1)                      # events the poly-A and the adaptor add to the first part
    lens = []
    for m in (1, 2, 3, 5):
        for d in (-2, -1, 0, 1, 2, 63, 64, 65):
            lens.append(512 * m - tail + d)
    lens = [n for n in lens if n >= k] * 3
    batches = [[bytes(rng.choice(list(b"ACGT"), n).astype(np.uint8)) for n in rng.permutation(lens)] for _ in range(2)]
    assert sum(len(r) + 237 for r in batches[0]) > 70000
    _check(prof, fl, k, 1, 99, batches, modes=(api.MODE_CERTIFIED,))
    _check(prof, fl, k, 2, 99, batches[:1], modes=(api.MODE_EXACT,))


@pytest.mark.gpu
@pytest.mark.parametrize("links", [None, "3000"])
@pytest.mark.parametrize("name,T", [("dna-r10-prom", 20), ("dna-r10-prom", 37), ("dna-r9-prom", 33)])
def test_many_worker_chains_each_with_several_reads(name, T, links, monkeypatch):
    """`-t 20 ... -t 37` with a dozen reads per worker: chains of very different weight, workers without a read in the second batch.
    By default such a batch has too few events per (worker chain, partition) for the bucketed hand-out and takes the per-link rows;
    forced links keep it there: more than 1024 pairs for the 9-mer table (k_part_slices walks them 1024 at a time)"""
    if links:
        monkeypatch.setenv("SQG_SPLIT_CHAINS", links)
    rng = np.random.default_rng(606 + T)
    prof, fl = profiles.get_profile(name)
    k = profiles.default_kmer_size(fl)
    batches = [_reads(rng, 12 * T, k, 1500), _reads(rng, T - 3, k, 1500) + _reads(rng, 5 * T, k, 900)]
    assert sum(max(len(r) - k + 1, 5) for r in batches[0]) > 70000
    _check(prof, fl, k, T, 5, batches, modes=(api.MODE_CERTIFIED,))


@pytest.mark.gpu
@pytest.mark.parametrize("name,T,lens", [("dna-r10-prom", 1, [180000]), ("dna-r9-prom", 1, [150000]), ("dna-r10-prom", 2, [90000, 70001]),
                                         ("rna004-prom", 3, [40000, 30000, 20000])])
def test_a_few_long_reads_are_cut_into_pieces(name, T, lens):
    """one read per worker, nothing but long reads: each is cut into pieces of whole segments that different wavefronts walk
    (k_part_events.h) -- the time stream's position, the tile offsets and the read's totals have to come out as in one walk.
    Two batches: the workers' streams carry over"""
    rng = np.random.default_rng(17)
    prof, fl = profiles.get_profile(name)
    if fl & profiles.SQ_RNA:
        fl |= profiles.SQ_PREFIX
    k = profiles.default_kmer_size(fl)
    batches = [[bytes(rng.choice(list(b"ACGT"), n).astype(np.uint8)) for n in lens] for _ in range(2)]
    _check(prof, fl, k, T, 3, batches, modes=(api.MODE_CERTIFIED,))


@pytest.mark.gpu
@pytest.mark.parametrize("knob", ["SQG_TEST_NO_LEAN", "SQG_TEST_DELTA_X"])
def test_queued_one_partition_batches_keep_their_own_events(knob, monkeypatch):
    """k <= 6, few workers (the one-partition hand-out): the sample kernels read rank and dwell from part[].  Batch i's generic
    kernel and fix-ups run on a stream of their own, next to batch i+1's event pass -- which must not write the part[] batch i is
    still reading (part[] is per slot).  Batches of different reads queued back to back, every tile through the slow kernels."""
    monkeypatch.setenv(knob, "1" if knob == "SQG_TEST_NO_LEAN" else "0.05")
    rng = np.random.default_rng(11)
    prof, fl = profiles.get_profile("dna-r9-prom")
    mean, stdv = model.synthetic_model(6)
    # a long batch (its slow kernels take a while) followed by short ones (their event passes are over quickly), several rounds
    sizes = [3000, 160, 160, 3000, 160]
    batches = [[bytes(rng.choice(list(b"ACGT"), int(m)).astype(np.uint8)) for m in rng.integers(300, 1500, n)] for n in sizes]
    orac = orc.Oracle(prof, fl, 6, mean, stdv, 42, num_workers=1)
    want = [orac.run_batch_seqs(bt, want_ss=False) for bt in batches]
    orac.close()
    gen = api.SignalGenerator(prof, fl, 6, mean, stdv, 42, num_workers=1, mode=api.MODE_CERTIFIED)
    # two batches in flight at any time (a batch's results live in its slot until the batch after the next one runs)
    staged = [gen.stage(bt) for bt in batches]
    staged[0].run()
    for bi in range(len(staged)):
        if bi + 1 < len(staged):
            staged[bi + 1].run()
        b = staged[bi].wait()
        sig = b.signal()
        for i, w in enumerate(want[bi]):
            np.testing.assert_array_equal(sig[b.sig_off[i]:b.sig_off[i + 1]], w.sig, err_msg=f"batch {bi} read {i}")
        b.free()
    gen.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["dna-r10-prom", "dna-r9-prom"])
def test_order_free_kernels_are_selectable_through_the_cfg(name):
    """SQG_ORDER_FREE in cfg.flags (not an environment variable): the context never relies on the lane order of LDS atomics and
    gives the same signals"""
    rng = np.random.default_rng(77)
    prof, fl = profiles.get_profile(name)
    k = profiles.default_kmer_size(fl)
    mean, stdv = model.synthetic_model(k)
    batches = [_reads(rng, 170, k, 3500) for _ in range(2)]
    out = []
    for extra in (0, profiles.SQ_ORDER_FREE):
        gen = api.SignalGenerator(prof, fl | extra, k, mean, stdv, 5, num_workers=1, mode=api.MODE_CERTIFIED)
        bad, used = gen.probe_lds_order(workgroups=64, rounds=4)
        assert bad == 0 and used == (extra == 0)
        sigs = []
        for bt in batches:
            b = gen.submit(bt)
            sigs.append(b.signal().copy())
            b.free()
        gen.close()
        out.append(sigs)
    for a, b in zip(*out):
        np.testing.assert_array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["dna-r10-prom", "dna-r9-prom"])
def test_every_batch_samples_the_lane_order_and_fails_loudly(name, monkeypatch):
    """k_part_hand_ord checks the first events of every slice against order-free prefix sums; with the atomics of the first two
    rows issued in the wrong order (test hook) the batch must fail, not return swapped streams"""
    rng = np.random.default_rng(78)
    prof, fl = profiles.get_profile(name)
    k = profiles.default_kmer_size(fl)
    mean, stdv = model.synthetic_model(k)
    reads = _reads(rng, 400, k, 3500)
    gen = api.SignalGenerator(prof, fl, k, mean, stdv, 5, num_workers=1, mode=api.MODE_CERTIFIED)
    b = gen.submit(reads)                                           # the healthy path passes its own check
    b.free()
    gen.close()
    monkeypatch.setenv("SQG_TEST_ORDER_FAULT", "1")
    gen = api.SignalGenerator(prof, fl, k, mean, stdv, 5, num_workers=1, mode=api.MODE_CERTIFIED)
    with pytest.raises(api.SqgError) as ei:
        gen.submit(reads)
    assert "lane order" in str(ei.value)
    gen.close()


@pytest.mark.gpu
@pytest.mark.parametrize("threads", [None, "2", "3", "7"])
@pytest.mark.parametrize("name,T", [("dna-r9-prom", 1), ("dna-r10-prom", 3), ("dna-r9-prom", 4), ("rna004-prom", 2)])
def test_thousands_of_reads_per_worker(name, T, threads, monkeypatch):
    """>= 4096 reads on <= 4 workers, two batches (carried streams): `offset`, `median_before` and every sample equal the oracle's serial
    walk -- also when the per-read draws of staging are shared by the context's helper threads (from 8192 reads per batch on;
    sqg_set_stage_threads fixes the number): a thread's range of the reads starts in the middle of a worker's chain, from the chain's
    streams moved past the reads before it (a read takes a fixed number of draws from each).  The call reports how many threads the
    last staging used: the helper path must actually have run."""
    rng = np.random.default_rng(4242 + T)
    prof, fl = profiles.get_profile(name)
    k = profiles.default_kmer_size(fl)
    mean, stdv = model.synthetic_model(k)
    batches = [[bytes(rng.choice(list(b"ACGT"), int(m)).astype(np.uint8)) for m in rng.integers(200, 420, n)] for n in (4500, 4100)]
    orac = orc.Oracle(prof, fl, k, mean, stdv, 42, num_workers=T)
    want = [orac.run_batch_seqs(bt, want_ss=False) for bt in batches]
    orac.close()
    gen = api.SignalGenerator(prof, fl, k, mean, stdv, 42, num_workers=T, mode=api.MODE_CERTIFIED)
    assert gen.set_stage_threads(int(threads) if threads else 0) == 0          # nothing staged yet
    for bi, bt in enumerate(batches):
        b = gen.submit(bt)
        used = gen.set_stage_threads(int(threads) if threads else 0)
        # (automatic: one below 8192 reads.  From the second batch on the context's draw-ahead thread may have made some of the draws
        # already: fewer are left to share)
        assert used == (int(threads) if threads else 1) if bi == 0 else 1 <= used <= (int(threads) if threads else 1)
        sig = b.signal()
        for i, w in enumerate(want[bi]):
            assert b.offset[i] == w.offset and b.median_before[i] == w.median_before, (bi, i)
            np.testing.assert_array_equal(sig[b.sig_off[i]:b.sig_off[i + 1]], w.sig, err_msg=f"batch {bi} read {i}")
        b.free()
    with pytest.raises(api.SqgError):
        gen.set_stage_threads(65)
    gen.close()


@pytest.mark.gpu
@pytest.mark.parametrize("every", [0, 2, 3])
def test_phase_timing_period_changes_the_timings_only(every):
    """sqg_set_phase_timing: batches without the phase events (recorded hipEvents between the kernels) report 0 ms and the same
    int16 as the batches that carry them; queued back to back, as a streaming caller does"""
    rng = np.random.default_rng(4242)
    prof, fl = profiles.get_profile("dna-r10-prom")
    k = profiles.default_kmer_size(fl)
    mean, stdv = model.synthetic_model(k)
    batches = [_reads(rng, 120, k, 3000) for _ in range(6)]
    sigs = {}
    for ev in (1, every):
        gen = api.SignalGenerator(prof, fl, k, mean, stdv, 11, num_workers=1, mode=api.MODE_CERTIFIED)
        gen.set_phase_timing(ev)
        out, timed = [], []
        staged = [gen.stage(bt) for bt in batches]
        for i in range(0, 6, 2):                         # two in flight
            staged[i].run(); staged[i + 1].run()
            for b in staged[i:i + 2]:
                b.wait()
                tm = gen.timing()
                timed.append(tm["total_ms"] > 0)
                assert (tm["lean_ms"] > 0) == timed[-1] and (tm["events_ms"] > 0) == timed[-1]
                assert tm["fallback_samples"] >= 0
                out.append(b.signal().copy())
        for b in staged:
            b.free()
        with pytest.raises(api.SqgError):
            gen.set_phase_timing(-1)
        gen.close()
        assert timed == [ev > 0 and i % ev == 0 for i in range(6)]
        sigs[ev] = out
    for a, b in zip(sigs[1], sigs[every]):
        np.testing.assert_array_equal(a, b)
