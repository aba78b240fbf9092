"""Range sharding ("strict -t 1" over several GPUs; include/sqg.h): every rank owns all workers
and generates a contiguous range of each batch's reads; the per-stream sample counts are the one exchange step.
Here G ranks are G contexts on one GPU and the exchange goes through the host; the result must be the oracle's
single-process run, read for read."""
import ctypes as C

import numpy as np
import pytest

import orc
from squigulator_amd import api, model, profiles
from test_sampler import NCOV, SEQUIN, _contigs


class Hip:
    """the three runtime calls the emulated exchange needs"""

    def __init__(self):
        self.L = C.CDLL("libamdhip64.so")
        self.L.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.L.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.L.hipFree.argtypes = [C.c_void_p]

    def to_host(self, dptr, n):
        out = np.empty(n, np.uint32)
        assert self.L.hipDeviceSynchronize() == 0
        assert self.L.hipMemcpy(out.ctypes.data, C.c_void_p(dptr), out.nbytes, 2) == 0
        return out

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr, np.uint32)
        p = C.c_void_p()
        assert self.L.hipMalloc(C.byref(p), max(arr.nbytes, 4)) == 0
        assert self.L.hipMemcpy(p, arr.ctypes.data, arr.nbytes, 1) == 0
        return p.value

    def free(self, p):
        self.L.hipFree(C.c_void_p(p))


def _ranges(n, G):
    cut = [n * g // G for g in range(G + 1)]
    return list(zip(cut[:-1], cut[1:]))


def _run(name, fasta, T, G, batches, rlen, sflags=0, mode=api.SAMPLE_DNA, seed=42, arith=api.MODE_CERTIFIED):
    hip = Hip()
    prof, fl = profiles.get_profile(name)
    k = profiles.default_kmer_size(fl)
    mean, stdv = model.synthetic_model(k)
    orac = orc.Oracle(prof, fl | sflags, k, mean, stdv, seed, num_workers=T, rlen=rlen)
    ref = orac.load_ref(fasta, None)
    gens = []
    for _ in range(G):
        g = api.SignalGenerator(prof, fl | sflags, k, mean, stdv, seed, num_workers=T, mode=arith)
        g.load_genome(_contigs(ref), rlen, mode)
        g.set_range_mode(True)
        gens.append(g)
    n_rows = T * (1 << (2 * k))
    for nb in batches:
        want = orac.run_batch(nb)
        rng = _ranges(nb, G)
        bs = [gens[g].sample(nb, lo=lo, hi=hi) for g, (lo, hi) in enumerate(rng)]
        counts = [hip.to_host(b.run_begin(), n_rows).astype(np.uint64) for b in bs]
        for g, b in enumerate(bs):
            before = sum(counts[:g], np.zeros(n_rows, np.uint64))
            after = sum(counts[g + 1:], np.zeros(n_rows, np.uint64))
            pb, pa = hip.to_device(before), hip.to_device(after)
            b.run_end(pb, pa).wait()
            hip.free(pb); hip.free(pa)
        for g, (b, (lo, hi)) in enumerate(zip(bs, rng)):
            sig, dw = b.signal(), b.dwell()
            s = b.sampled
            for i in range(hi - lo):
                w = want[lo + i]
                assert (s["ref_idx"][i], s["ref_pos"][i], s["rlen"][i], chr(s["strand"][i])) == (w.ref_idx, w.ref_pos_st, w.rlen, w.strand)
                np.testing.assert_array_equal(dw[b.ev_off[i]:b.ev_off[i + 1]], w.ss, err_msg=f"rank {g} read {lo + i}")
                np.testing.assert_array_equal(sig[b.sig_off[i]:b.sig_off[i + 1]], w.sig, err_msg=f"rank {g} read {lo + i}")
                assert b.offset[i] == w.offset and b.median_before[i] == w.median_before
            b.free()
    for g in gens:
        g.close()
    orac.close()


@pytest.mark.gpu
@pytest.mark.parametrize("T,G,batches", [(1, 2, [9, 20, 4]), (1, 3, [30, 2, 17]), (2, 3, [12, 25]), (3, 2, [7, 7])],
                         ids=["t1_g2", "t1_g3", "t2_g3", "t3_g2"])
def test_dna_r9_ranges_equal_the_single_process_run(T, G, batches):
    _run("dna-r9-prom", NCOV, T, G, batches, rlen=900)


@pytest.mark.gpu
def test_nine_mer_counts_are_exchanged_as_sums():
    _run("dna-r10-prom", NCOV, 1, 2, [10, 14], rlen=700)


@pytest.mark.gpu
def test_rna_prefix_and_exact_mode():
    _run("rna004-prom", SEQUIN, 1, 2, [8, 9], rlen=10000, sflags=profiles.SQ_PREFIX, mode=api.SAMPLE_RNA)
    _run("dna-r9-prom", NCOV, 1, 2, [8, 9], rlen=600, arith=api.MODE_EXACT)


@pytest.mark.gpu
def test_a_rank_with_an_empty_range_still_follows_the_streams():
    """3 ranks, 2 reads per batch: one range is empty every time"""
    _run("dna-r9-prom", NCOV, 1, 3, [2, 2, 5], rlen=500)


@pytest.mark.gpu
def test_host_reads_with_skip_reads_match_the_oracle():
    """reads staged from the host: the ranks call sqg_skip_reads for the reads before and after their range"""
    hip = Hip()
    rng_ = np.random.default_rng(3)
    prof, fl = profiles.get_profile("dna-r9-prom")
    mean, stdv = model.synthetic_model(6)
    T, G = 2, 2
    batches = [[bytes(rng_.choice(list(b"ACGT"), int(m)).astype(np.uint8)) for m in rng_.integers(5, 900, nb)] for nb in (11, 6)]
    orac = orc.Oracle(prof, fl, 6, mean, stdv, 5, num_workers=T)
    want = [orac.run_batch_seqs(bt) for bt in batches]
    orac.close()
    gens = [api.SignalGenerator(prof, fl, 6, mean, stdv, 5, num_workers=T, mode=api.MODE_CERTIFIED) for _ in range(G)]
    for g in gens:
        g.set_range_mode(True)
    n_rows = T * 4096
    for bi, bt in enumerate(batches):
        n = len(bt)
        wk = np.array([gens[0].L.sqg_worker_of(i, n, T) for i in range(n)], np.int32)
        lens = np.array([len(r) for r in bt], np.int64)
        rng = _ranges(n, G)
        bs = []
        for g, (lo, hi) in enumerate(rng):
            gens[g].skip_reads(lens[:lo], wk[:lo])
            bs.append(gens[g].stage(bt[lo:hi], wk[lo:hi]))
            gens[g].skip_reads(lens[hi:], wk[hi:])
        counts = [hip.to_host(b.run_begin(), n_rows).astype(np.uint64) for b in bs]
        for g, b in enumerate(bs):
            pb = hip.to_device(sum(counts[:g], np.zeros(n_rows, np.uint64)))
            pa = hip.to_device(sum(counts[g + 1:], np.zeros(n_rows, np.uint64)))
            b.run_end(pb, pa).wait()
            hip.free(pb); hip.free(pa)
            sig = b.signal()
            lo = rng[g][0]
            for i in range(b.n_reads):
                np.testing.assert_array_equal(sig[b.sig_off[i]:b.sig_off[i + 1]], want[bi][lo + i].sig, err_msg=f"batch {bi} read {lo + i}")
                assert b.offset[i] == want[bi][lo + i].offset
            b.free()
    for g in gens:
        g.close()


@pytest.mark.gpu
def test_two_phase_run_misuse_is_reported():
    prof, fl = profiles.get_profile("dna-r9-prom")
    mean, stdv = model.synthetic_model(6)
    gen = api.SignalGenerator(prof, fl, 6, mean, stdv, 1, num_workers=1, mode=api.MODE_CERTIFIED)
    reads = [b"ACGT" * 100] * 3
    b = gen.stage(reads)
    with pytest.raises(api.SqgError):
        b.run_begin()                                    # range mode is off
    with pytest.raises(api.SqgError):
        b.run_end()                                      # nothing begun
    with pytest.raises(api.SqgError):
        gen.set_range_mode(True)                         # a staged batch is pending
    b.run().wait(); b.free()
    gen.set_range_mode(True)
    b = gen.stage(reads)
    b.run_begin()
    with pytest.raises(api.SqgError):
        b.run_begin()                                    # twice
    with pytest.raises(api.SqgError):
        b.run()                                          # begun: only run_end may follow
    with pytest.raises(api.SqgError):
        b.run_end(1, None)                               # before without after
    b.run_end().wait()
    want = b.signal().copy()
    b.free()
    # range mode with the whole batch here == the plain run of a fresh context
    gen2 = api.SignalGenerator(prof, fl, 6, mean, stdv, 1, num_workers=1, mode=api.MODE_CERTIFIED)
    x = gen2.submit(reads); x.free()
    y = gen2.submit(reads)
    np.testing.assert_array_equal(y.signal(), want)
    y.free(); gen2.close(); gen.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(8))
def test_random_worker_counts_rank_counts_and_batch_sizes(seed):
    rng = np.random.default_rng(4000 + seed)
    name, fasta, mode, sflags, rlen = [("dna-r9-prom", NCOV, api.SAMPLE_DNA, 0, 700), ("dna-r10-prom", NCOV, api.SAMPLE_DNA, 0, 500),
                                      ("rna004-prom", SEQUIN, api.SAMPLE_RNA, profiles.SQ_PREFIX, 10000),
                                      ("dna-r9-prom", NCOV, api.SAMPLE_DNA, profiles.SQ_PREFIX, 400)][seed % 4]
    T, G = int(rng.integers(1, 5)), int(rng.integers(2, 5))
    batches = [int(rng.integers(1, 24)) for _ in range(3)]
    _run(name, fasta, T, G, batches, rlen, sflags=sflags, mode=mode, seed=int(rng.integers(1, 1 << 20)))


@pytest.mark.gpu
def test_results_of_a_reused_slot_are_refused():
    """Outputs live in two slots used alternately.  In range mode sqg_batch_run_begin of batch r+2 already writes the slot
    of batch r (dwells, lengths): from then on batch r's device results are gone -- sqg_fetch_* must say so instead of
    handing out the newer batch's data, and sqg_batch_wait must not return device pointers into it."""
    prof, fl = profiles.get_profile("dna-r9-prom")
    mean, stdv = model.synthetic_model(6)
    g = api.SignalGenerator(prof, fl, 6, mean, stdv, 42, num_workers=1, mode=api.MODE_CERTIFIED)
    rng = np.random.default_rng(3)
    hip = Hip()
    zero = hip.to_device(np.zeros(4096, np.uint32))
    g.set_range_mode(True)
    bs = []
    for _ in range(3):
        bs.append(g.stage([bytes(rng.choice(list(b"ACGT"), 700).astype(np.uint8)) for _ in range(6)]))
    for b in bs[:2]:
        b.run_begin()
        b.run_end(zero, zero).wait()
    d0 = bs[0].dwell()                                     # still there
    assert len(d0) == bs[0].n_events
    bs[2].run_begin()                                      # takes the slot of batch 0
    with pytest.raises(api.SqgError) as e:
        bs[0].dwell()
    assert e.value.code == -4
    with pytest.raises(api.SqgError):
        bs[0].signal()
    bs[0].wait()
    assert not bs[0].res.d_signal and not bs[0].res.d_dwell
    assert len(bs[1].dwell()) == bs[1].n_events            # the other slot is untouched
    bs[2].run_end(zero, zero).wait()
    np.testing.assert_array_equal(bs[2].dwell() > 0, True)
    for b in bs:
        b.free()
    hip.free(zero)
    g.close()
