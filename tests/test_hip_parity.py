"""GPU parity: the HIP path, called through the C ABI, must reproduce
  (a) the committed vectors of the COMPILED REFERENCE path (tests/golden/refvec), and
  (b) the oracle on fresh seeded inputs,
bit for bit (int16 samples, per-event dwell, len_raw_signal, offset, median_before)."""
import os

import numpy as np
import pytest

import hiprun
import simrun
from refvec_cases import REFVEC_CASES
from squigulator_amd import api

pytestmark = pytest.mark.gpu

VEC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "refvec")


def _fixture_reads(v):
    meta = v["meta"]
    so = go = eo = 0
    out = []
    for i in range(len(meta)):
        rlen, nsig, nss = int(meta[i][4]), int(meta[i][7]), int(meta[i][8])
        out.append(dict(seq=v["seq"][so:so + rlen].tobytes(), sig=v["sig"][go:go + nsig], ss=v["ss"][eo:eo + nss],
                        offset=float(v["offset"][i]), median=float(v["median"][i]), start_time=int(meta[i][6])))
        so += rlen; go += nsig; eo += nss
    return out


def _compare(got, want, what):
    assert len(got) == len(want)
    for i, (g, w) in enumerate(zip(got, want)):
        assert len(g["sig"]) == len(w["sig"]), f"{what} read {i}: len_raw_signal {len(g['sig'])} != {len(w['sig'])}"
        np.testing.assert_array_equal(g["ss"], w["ss"], err_msg=f"{what} read {i}: per-event dwell")
        bad = np.nonzero(g["sig"] != w["sig"])[0]
        assert bad.size == 0, f"{what} read {i}: {bad.size} samples differ, first at {bad[:5]}: {g['sig'][bad[:5]]} vs {w['sig'][bad[:5]]}"
        assert g["offset"] == w["offset"] and g["median"] == w["median"], f"{what} read {i}: offset/median_before"
        assert g["start_time"] == w["start_time"]


@pytest.mark.parametrize("mode", [api.MODE_EXACT, api.MODE_CERTIFIED], ids=["exact", "certified"])
@pytest.mark.parametrize("cid,cmd", REFVEC_CASES, ids=[c[0] for c in REFVEC_CASES])
def test_hip_matches_reference_vectors(cid, cmd, mode):
    v = np.load(os.path.join(VEC, cid + ".npz"))
    want = _fixture_reads(v)
    got = hiprun.run_hip_on_reads(cmd, [w["seq"] for w in want], mode=mode)
    _compare(got, want, cid)


@pytest.mark.parametrize("cmd", [
    "nCoV-2019.reference.fasta -x dna-r9-prom -n 300 --seed 42 -r 3000 -t 128 -K 128",
    "nCoV-2019.reference.fasta -x dna-r9-prom -n 12 --seed 9 -r 8000 -t 5 -K 12",     # 1<T<K static partition
    "nCoV-2019.reference.fasta -x dna-r10-prom -n 64 --seed 5 -r 2000 -t 32 -K 32",
    "rnasequin_sequences_2.4.fa -x rna-r9-min -n 20 --seed 11 -t 20 -K 20 --prefix=yes",
    "rnasequin_sequences_2.4.fa -x rna004-prom -n 24 --seed 13 -t 8 -K 8 --prefix=yes --dwell-std 5",
    "nCoV-2019.reference.fasta -x dna-r9-min -n 6 --seed 3 -r 1000 -t1 --dwell-mean 20 --dwell-std 30",
    # CpG methylation: 5^6 = 15625 streams per worker -- one worker per read (rows in HBM), and -t 1 / -t 3 with enough events
    # for the chains to be cut (bucketed hand-out over 4 partitions)
    "nCoV-2019.reference.fasta -x dna-r9-prom -n 48 --seed 5 -r 2500 -t 24 -K 24 --meth-freq mfreq_dense.tsv",
    "nCoV-2019.reference.fasta -x dna-r9-prom -n 70 --seed 6 -r 2500 -t 1 -K 40 --meth-freq mfreq_dense.tsv",
    "nCoV-2019.reference.fasta -x dna-r9-prom -n 60 --seed 7 -r 2500 -t 3 -K 60 --meth-freq mfreq.tsv --prefix=yes",
], ids=["r9_tk128", "r9_t5_k12", "r10_tk32", "rna9min_prefix", "rna004_prefix_dwellstd", "r9_wide_dwell",
        "meth_tk24", "meth_t1", "meth_t3_prefix"])
@pytest.mark.parametrize("mode", [api.MODE_EXACT, api.MODE_CERTIFIED], ids=["exact", "certified"])
def test_hip_matches_oracle(cmd, mode):
    o, k, names, lengths, reads, orac = simrun.run_oracle(cmd, nthreads=8)
    orac.close()
    want = [dict(seq=r.seq, sig=r.sig, ss=r.ss, offset=r.offset, median=r.median_before, start_time=r.start_time)
            for r in reads]
    got = hiprun.run_hip_on_reads(cmd, [w["seq"] for w in want], mode=mode)
    _compare(got, want, cmd)


@pytest.mark.parametrize("delta", ["1.0", "3e-4"], ids=["all_fp64", "many_fp64"])
def test_certified_fallback_branch_is_exact(delta, monkeypatch):
    """The rare FP64 fix-up branch, FORCED: with an inflated error bound every (or ~1%) sample is
    rejected by the fp32 acceptance test and recomputed by k_fixup; output must not change."""
    cmd = "rnasequin_sequences_2.4.fa -x rna004-prom -n 12 --seed 21 -t 12 -K 12 --prefix=yes --dwell-std 4"
    o, k, names, lengths, reads, orac = simrun.run_oracle(cmd, nthreads=8)
    orac.close()
    want = [dict(seq=r.seq, sig=r.sig, ss=r.ss, offset=r.offset, median=r.median_before, start_time=r.start_time)
            for r in reads]
    monkeypatch.setenv("SQG_TEST_DELTA_X", delta)
    got = hiprun.run_hip_on_reads(cmd, [w["seq"] for w in want], mode=api.MODE_CERTIFIED)
    _compare(got, want, cmd)
    assert hiprun.LAST_FALLBACK > (0.5 if delta == "1.0" else 0.001) * sum(len(w["sig"]) for w in want)


def test_certified_fallback_rate_is_small():
    cmd = "nCoV-2019.reference.fasta -x dna-r9-prom -n 64 --seed 4 -r 4000 -t 64 -K 64"
    o, k, names, lengths, reads, orac = simrun.run_oracle(cmd, nthreads=8)
    orac.close()
    want = [dict(seq=r.seq, sig=r.sig, ss=r.ss, offset=r.offset, median=r.median_before, start_time=r.start_time)
            for r in reads]
    got = hiprun.run_hip_on_reads(cmd, [w["seq"] for w in want], mode=api.MODE_CERTIFIED)
    _compare(got, want, cmd)
    total = sum(len(w["sig"]) for w in want)
    assert 0 < hiprun.LAST_FALLBACK < 2e-3 * total, (hiprun.LAST_FALLBACK, total)


def test_short_and_odd_reads():
    """reads shorter than k (src/gensig.c:242-245), IUPAC/lower-case bases, a 1-event read, empty batch."""
    import orc
    from squigulator_amd import model, profiles
    prof, fl = profiles.get_profile("dna-r9-prom")
    mean, stdv = model.synthetic_model(6)
    seqs = [b"ACG", b"ACGTAC", b"acgtnnRYKMacgtacgtBDHVacgtuUwWsS", b"A" * 50, b"ACGTACGTACGTACGTTTGACCA" * 20]
    orac = orc.Oracle(prof, fl, 6, mean, stdv, 77, num_workers=len(seqs))
    want = orac.run_batch_seqs(seqs)
    gen = api.SignalGenerator(prof, fl, 6, mean, stdv, 77, num_workers=len(seqs))
    b = gen.submit(seqs)
    sig, dw = b.signal(), b.dwell()
    for i, w in enumerate(want):
        np.testing.assert_array_equal(sig[b.sig_off[i]:b.sig_off[i + 1]], w.sig, err_msg=f"read {i}")
        np.testing.assert_array_equal(dw[b.ev_off[i]:b.ev_off[i + 1]], w.ss)
        assert b.offset[i] == w.offset and b.median_before[i] == w.median_before
    b.free()
    e = gen.submit([])
    assert e.n_samples == 0 and e.n_reads == 0
    e.free()
    gen.close(); orac.close()


def test_worker_shards_equal_one_context():
    """Multi-GPU sharding at the C ABI: two contexts owning workers [0,5) and [5,12) (as two GPUs would)
    reproduce, read for read, what one context owning all 12 workers produces -- over two batches."""
    import orc
    from squigulator_amd import model, profiles, shard
    prof, fl = profiles.get_profile("dna-r10-prom")
    k = 9
    mean, stdv = model.synthetic_model(k)
    rng = np.random.default_rng(5)
    T = 12
    batches = [[bytes(rng.choice(list(b"ACGT"), size=int(n)).astype(np.uint8)) for n in rng.integers(300, 1500, size=T)]
               for _ in range(2)]
    whole = api.SignalGenerator(prof, fl, k, mean, stdv, seed=9, num_workers=T, mode=api.MODE_CERTIFIED)
    parts = [api.SignalGenerator(prof, fl, k, mean, stdv, seed=9, num_workers=T, mode=api.MODE_CERTIFIED,
                                 worker_lo=lo, worker_hi=hi) for lo, hi in ((0, 5), (5, 12))]
    for reads in batches:
        bw = whole.submit(reads)
        sw = bw.signal()
        for g, (lo, hi) in zip(parts, ((0, 5), (5, 12))):
            idx = [i for i in range(T) if lo <= i < hi]
            bp = g.submit([reads[i] for i in idx], workers=idx)
            sp = bp.signal()
            for j, i in enumerate(idx):
                np.testing.assert_array_equal(sp[bp.sig_off[j]:bp.sig_off[j + 1]], sw[bw.sig_off[i]:bw.sig_off[i + 1]])
                assert bp.offset[j] == bw.offset[i] and bp.median_before[j] == bw.median_before[i]
            bp.free()
        bw.free()
    with pytest.raises(api.SqgError):
        parts[0].submit([batches[0][0]], workers=[7])          # a worker this context does not own
    whole.close()
    for g in parts:
        g.close()


def test_paf_sam_goldens_from_device_dwell():
    """the per-event dwell array of the HIP path feeds PAF/SAM `ss:Z:` emission; with the
    reads of the reference's own golden runs (scripts/test.sh:81-99) the product's writers reproduce
    test/dna_r10_paf.paf.exp and test/rna_paf.{paf,sam}.exp byte for byte (these files do not depend on the
    absent pore-model tables)."""
    import gzip
    from refcases import CASES
    from squigulator_amd import aln_text, slow5_text
    exp_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_exp")
    for cid in ("r10_paf", "rna_paf_sam", "r10_paf_ref"):
        _, cmd, exp = next(c for c in CASES if c[0] == cid)
        o, k, names, lengths, reads, orac = simrun.run_oracle(cmd)      # the oracle only SAMPLES the reads here
        orac.close()
        got = hiprun.run_hip_on_reads(cmd, [r.seq for r in reads], mode=api.MODE_CERTIFIED)
        paf, sam = [], [aln_text.sam_header(names, lengths)]
        for r, g in zip(reads, got):
            rid = slow5_text.read_id(o.flags, r.read_number + 1, names[r.ref_idx], r.ref_pos_st, r.ref_pos_end, r.strand)
            a = aln_text.Aln(o.flags, k, rid, names[r.ref_idx], r.ref_len, r.ref_pos_st, r.ref_pos_end, r.strand,
                             r.rlen, len(g["sig"]), g["ss"])
            paf.append(aln_text.paf_str(a))
            sam.append(aln_text.sam_str(a, r.seq.decode(), names[r.ref_idx], r.ref_pos_st))
        for kind, text in (("paf", "".join(paf)), ("sam", "".join(sam))):
            if kind in exp:
                with gzip.open(os.path.join(exp_dir, exp[kind] + ".gz"), "rt") as f:
                    assert text == f.read(), f"{cid}: {kind}"


@pytest.mark.gpu
def test_batches_queued_back_to_back_keep_their_results():
    """Double-buffered slots: batch i can be fetched after batch i+1 has been run (not after i+2), batches run without
    intermediate waits equal the oracle, and waiting out of order is fine."""
    import orc
    from squigulator_amd import model, profiles
    prof, fl = profiles.get_profile("dna-r9-prom")
    mean, stdv = model.synthetic_model(6)
    rng = np.random.default_rng(3)
    T = 6
    batches = [[bytes(rng.choice(list(b"ACGT"), int(n)).astype(np.uint8)) for n in rng.integers(30, 900, T)] for _ in range(4)]
    orac = orc.Oracle(prof, fl, 6, mean, stdv, 99, num_workers=T)
    want = [orac.run_batch_seqs(bt) for bt in batches]
    gen = api.SignalGenerator(prof, fl, 6, mean, stdv, 99, num_workers=T, mode=api.MODE_CERTIFIED)
    hs = [gen.stage(bt) for bt in batches]
    hs[0].run(); hs[1].run()                       # two in flight
    hs[1].wait(); hs[0].wait()                     # out of order
    s0, s1 = hs[0].signal(), hs[1].signal()        # batch 0 is still there after batch 1 ran
    hs[2].run(); hs[3].run()
    with pytest.raises(api.SqgError) as e:         # ... but not after batch 2 reused its slot
        hs[0].signal()
    assert e.value.code == -4
    hs[3].wait(); hs[2].wait()
    sigs = [s0, s1, hs[2].signal(), hs[3].signal()]
    for bi in range(4):
        for i, w in enumerate(want[bi]):
            np.testing.assert_array_equal(sigs[bi][hs[bi].sig_off[i]:hs[bi].sig_off[i + 1]], w.sig, err_msg=f"batch {bi} read {i}")
    for h in hs:
        h.free()
    gen.close(); orac.close()


@pytest.mark.gpu
def test_a_staged_batch_freed_without_a_run_does_not_block_the_ones_behind_it():
    from squigulator_amd import model, profiles
    rng = np.random.default_rng(8)
    prof, fl = profiles.get_profile("dna-r9-prom")
    mean, stdv = model.synthetic_model(6)
    gen = api.SignalGenerator(prof, fl, 6, mean, stdv, 3, num_workers=4, mode=api.MODE_CERTIFIED)
    reads = [bytes(rng.choice(list(b"ACGT"), 300).astype(np.uint8)) for _ in range(4)]
    a, b, c = gen.stage(reads), gen.stage(reads), gen.stage(reads)
    b.free()                                             # never run
    a.run()
    c.run()                                              # would be out of sequence if b still held its place
    a.wait(); c.wait()
    sa, sc = a.signal().copy(), c.signal().copy()        # a's results are still there: c ran in the other buffer set
    assert len(sa) == a.n_samples and len(sc) == c.n_samples
    d = gen.stage(reads).run().wait()
    assert len(d.signal()) == d.n_samples
    with pytest.raises(api.SqgError):
        a.signal()                                       # two batches have run since: a's slab has been reused
    for x in (a, c, d):
        x.free()
    gen.close()
