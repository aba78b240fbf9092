"""Device-side read sampler on the resident genome ("next" row): gen_read, src/genread.c:125-370.

The oracle's sampler is pinned by the reference's goldens (read ids = contig/position/strand of every read,
tests/test_oracle_goldens.py); here the device sampler must reproduce it read for read -- coordinates, the
sequence after N substitution and reverse complement, and through them the signals."""
import os

import numpy as np
import pytest

import orc
from squigulator_amd import api, model, profiles

INPUTS = os.path.join(os.path.dirname(__file__), "golden", "inputs")
NCOV = os.path.join(INPUTS, "nCoV-2019.reference.fasta")
SEQUIN = os.path.join(INPUTS, "rnasequin_sequences_2.4.fa")
SEQUIN_TC = os.path.join(INPUTS, "sequin_count.tsv")


def _contigs(ref):
    return [bytes(ref.seqs[i][:ref.lengths[i]]) for i in range(ref.num_ref)]


def _run(profile, k, fasta, T, batches, rlen, oflags=0, sflags=0, mode=api.SAMPLE_DNA, trans_count=None, seed=42, meth_freq=None):
    prof, fl = profiles.get_profile(profile)
    if meth_freq:
        sflags |= profiles.SQ_METH
    mean, stdv = model.synthetic_model(k, meth=bool(meth_freq))
    orac = orc.Oracle(prof, fl | oflags | sflags, k, mean, stdv, seed, num_workers=T, rlen=rlen)
    ref = orac.load_ref(fasta, trans_count, meth_freq)
    trans = None
    if trans_count:
        trans = (np.ctypeslib.as_array(ref.trans_csum, shape=(ref.trans_n,)).copy(),
                 np.ctypeslib.as_array(ref.trans_idx, shape=(ref.trans_n,)).copy())
    gen = api.SignalGenerator(prof, fl | sflags, k, mean, stdv, seed, num_workers=T, mode=api.MODE_CERTIFIED)
    gen.load_genome(_contigs(ref), rlen, mode, trans)
    if meth_freq:
        gen.set_meth(_contigs(ref), [ref.names[i].decode() for i in range(ref.num_ref)], meth_freq)
    total = 0
    for nb in batches:
        want = orac.run_batch(nb)
        b = gen.sample(nb).run().wait()
        s = b.sampled
        seqs = b.reads()
        sig, dw = b.signal(), b.dwell()
        for i, w in enumerate(want):
            assert (s["ref_idx"][i], s["ref_pos"][i], s["rlen"][i], chr(s["strand"][i])) == \
                   (w.ref_idx, w.ref_pos_st, w.rlen, w.strand), f"read {i}"
            assert s["ref_len"][i] == w.ref_len
            assert seqs[i] == w.seq, f"read {i}: sequence differs"
            np.testing.assert_array_equal(sig[b.sig_off[i]:b.sig_off[i + 1]], w.sig, err_msg=f"read {i}")
            np.testing.assert_array_equal(dw[b.ev_off[i]:b.ev_off[i + 1]], w.ss)
        total += nb
        b.free()
    gen.close(); orac.close()
    return total


@pytest.mark.gpu
@pytest.mark.parametrize("T,batches", [(1, [12, 5]), (16, [16, 16, 7]), (4, [10])], ids=["t1", "tk16", "t4_k10"])
def test_dna_sampler_matches_oracle(T, batches):
    _run("dna-r9-prom", 6, NCOV, T, batches, rlen=1500)


@pytest.mark.gpu
def test_dna_sampler_with_N_runs_and_lowercase(tmp_path):
    """N substitution from the fresh state-100 stream, >10 % N rejection, clipping at contig ends, lower case."""
    rng = np.random.default_rng(11)
    contigs = []
    for n in (3000, 9000, 1200, 300):
        s = rng.choice(list(b"ACGTacgt"), n).astype(np.uint8)
        for _ in range(n // 400):
            p = int(rng.integers(0, n - 60))
            s[p:p + int(rng.integers(1, 60))] = ord("N")
        contigs.append(bytes(s))
    contigs[1] = contigs[1][:4000] + b"N" * 900 + contigs[1][4900:]
    fa = tmp_path / "g.fa"
    fa.write_text("".join(f">c{i}\n{c.decode()}\n" for i, c in enumerate(contigs)))
    _run("dna-r9-prom", 6, str(fa), 8, [8, 8, 8], rlen=900, seed=7)


@pytest.mark.gpu
@pytest.mark.parametrize("name,oflags,mode,tc", [
    ("uniform", 0, api.SAMPLE_RNA, None),
    ("trans_count", 0, api.SAMPLE_RNA, SEQUIN_TC),
    ("trans_trunc", 0x100, api.SAMPLE_RNA | api.SAMPLE_TRUNC, SEQUIN_TC),
], ids=lambda x: x if isinstance(x, str) else None)
def test_rna_sampler_matches_oracle(name, oflags, mode, tc):
    _run("rna004-prom", 9, SEQUIN, 6, [6, 6], rlen=10000, oflags=oflags, sflags=profiles.SQ_PREFIX, mode=mode, trans_count=tc)


@pytest.mark.gpu
def test_cdna_sampler_matches_oracle():
    _run("dna-r9-prom", 6, SEQUIN, 5, [5, 5], rlen=10000, oflags=0x200, mode=api.SAMPLE_CDNA, trans_count=SEQUIN_TC)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(10))
def test_random_genomes(seed, tmp_path):
    """Random contig sets (tiny contigs that force clipping and <200-nt rejections, N runs, lower case), read lengths,
    worker counts and batch sizes."""
    rng = np.random.default_rng(500 + seed)
    contigs = []
    for _ in range(int(rng.integers(1, 12))):
        n = int(rng.choice([210, 260, 400, 1500, 6000, 20000]))
        a = rng.choice(list(b"ACGTacgtRYN"), n, p=[.23, .23, .23, .23, .01, .01, .01, .01, .01, .01, .02]).astype(np.uint8)
        if rng.random() < 0.5:
            q = int(rng.integers(0, max(n - 50, 1)))
            a[q:q + int(rng.integers(5, 120))] = ord("N")
        contigs.append(bytes(a))
    fa = tmp_path / "g.fa"
    fa.write_text("".join(f">c{i}\n{c.decode()}\n" for i, c in enumerate(contigs)))
    T = int(rng.integers(1, 9))
    batches = [int(rng.integers(1, 2 * T + 1)) for _ in range(3)]
    _run("dna-r9-prom", 6, str(fa), T, batches, rlen=int(rng.choice([300, 1000, 5000])), seed=int(rng.integers(1, 1 << 20)))


@pytest.mark.gpu
@pytest.mark.parametrize("T,batches", [(1, [40, 25, 16]), (3, [100, 64])], ids=["t1", "t3"])
def test_long_chains_are_sampled_concurrently_and_match_the_oracle(T, batches):
    """>= 16 reads per worker: the attempts are evaluated side by side (k_sample_try / k_sample_pick)"""
    _run("dna-r9-prom", 6, NCOV, T, batches, rlen=800)


@pytest.mark.gpu
@pytest.mark.parametrize("name,oflags,mode,tc", [
    ("uniform", 0, api.SAMPLE_RNA, None),
    ("trans_trunc", 0x100, api.SAMPLE_RNA | api.SAMPLE_TRUNC, SEQUIN_TC),
    ("cdna", 0x200, api.SAMPLE_CDNA, SEQUIN_TC),
], ids=lambda x: x if isinstance(x, str) else None)
def test_long_chains_rna_variants(name, oflags, mode, tc):
    if mode & api.SAMPLE_CDNA:
        _run("dna-r9-prom", 6, SEQUIN, 2, [40, 33], rlen=10000, oflags=oflags, mode=mode, trans_count=tc)
    else:
        _run("rna004-prom", 9, SEQUIN, 2, [40, 33], rlen=10000, oflags=oflags, sflags=profiles.SQ_PREFIX, mode=mode, trans_count=tc)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(6))
def test_long_chains_on_genomes_that_reject_most_attempts(seed, tmp_path):
    """tiny contigs and N runs: most attempts are rejected, so the attempt slots run out and the chain is finished one
    read at a time; the next batch sizes its slots by the rate seen"""
    rng = np.random.default_rng(900 + seed)
    contigs = []
    for _ in range(int(rng.integers(2, 9))):
        n = int(rng.choice([205, 230, 300, 900, 4000]))
        a = rng.choice(list(b"ACGTN"), n, p=[.24, .24, .24, .24, .04]).astype(np.uint8)
        q = int(rng.integers(0, max(n - 50, 1)))
        a[q:q + int(rng.integers(5, 60))] = ord("N")
        contigs.append(bytes(a))
    fa = tmp_path / "g.fa"
    fa.write_text("".join(f">c{i}\n{c.decode()}\n" for i, c in enumerate(contigs)))
    T = int(rng.integers(1, 4))
    _run("dna-r9-prom", 6, str(fa), T, [int(rng.integers(16 * T, 30 * T)) for _ in range(3)], rlen=int(rng.choice([250, 600])),
         seed=int(rng.integers(1, 1 << 20)))


@pytest.mark.gpu
def test_full_contigs_are_handed_out_in_order_and_as_loaded(tmp_path):
    """--full-contigs (src/sim.c:543-549): read i of the job is contig i; N and lower case stay as they are; asking for
    more reads than contigs is an error"""
    _run("dna-r9-prom", 6, NCOV, 1, [1], rlen=10000, oflags=0x002, mode=api.SAMPLE_FULL)
    rng = np.random.default_rng(4)
    contigs = [bytes(rng.choice(list(b"ACGTacgtNRY"), n, p=[.2, .2, .2, .2, .04, .04, .04, .04, .02, .01, .01]).astype(np.uint8))
               for n in (700, 5, 2600, 64, 1300, 9000, 3)]
    fa = tmp_path / "g.fa"
    fa.write_text("".join(f">c{i}\n{c.decode()}\n" for i, c in enumerate(contigs)))
    _run("dna-r9-prom", 6, str(fa), 3, [3, 2, 2], rlen=10000, oflags=0x002, mode=api.SAMPLE_FULL)
    prof, fl = profiles.get_profile("dna-r9-prom")
    mean, stdv = model.synthetic_model(6)
    gen = api.SignalGenerator(prof, fl, 6, mean, stdv, 1, num_workers=2, mode=api.MODE_CERTIFIED)
    gen.load_genome(contigs, 10000, api.SAMPLE_FULL)
    gen.sample(5).free()
    with pytest.raises(api.SqgError):
        gen.sample(3)                                    # two contigs left
    gen.close()


@pytest.mark.gpu
def test_sampler_shards_equal_one_context():
    """Multi-GPU sharding of the sampler: a context that owns workers [lo, hi) draws exactly the reads the
    single-context run draws for those workers (streams are per worker; no exchange)."""
    prof, fl = profiles.get_profile("dna-r9-prom")
    mean, stdv = model.synthetic_model(6)
    orac = orc.Oracle(prof, fl, 6, mean, stdv, 5, num_workers=1)
    contigs = _contigs(orac.load_ref(NCOV))
    orac.close()
    T = 12
    one = api.SignalGenerator(prof, fl, 6, mean, stdv, 5, num_workers=T, mode=api.MODE_CERTIFIED)
    one.load_genome(contigs, 1200)
    shards = []
    for lo, hi in ((0, 5), (5, 12)):
        g = api.SignalGenerator(prof, fl, 6, mean, stdv, 5, num_workers=T, mode=api.MODE_CERTIFIED, worker_lo=lo, worker_hi=hi)
        g.load_genome(contigs, 1200)
        shards.append((lo, hi, g))
    for _ in range(3):                                         # several batches: the streams carry over
        b = one.sample(T).run().wait()
        sig = b.signal()
        for lo, hi, g in shards:
            bs = g.sample(hi - lo, workers=np.arange(lo, hi, dtype=np.int32)).run().wait()
            ss = bs.signal()
            for j in range(hi - lo):
                i = lo + j
                for key in ("ref_idx", "ref_pos", "rlen"):
                    assert bs.sampled[key][j] == b.sampled[key][i]
                assert bs.sampled["strand"][j] == b.sampled["strand"][i]
                np.testing.assert_array_equal(ss[bs.sig_off[j]:bs.sig_off[j + 1]], sig[b.sig_off[i]:b.sig_off[i + 1]])
            bs.free()
        b.free()
    one.close()
    for _, _, g in shards:
        g.close()


@pytest.mark.gpu
def test_sampler_and_compress_error_paths():
    """Loud failures, no silent fallbacks: sampling without a genome, bad genome arguments, stale batches."""
    prof, fl = profiles.get_profile("dna-r9-prom")
    mean, stdv = model.synthetic_model(6)
    gen = api.SignalGenerator(prof, fl, 6, mean, stdv, 1, num_workers=2, mode=api.MODE_CERTIFIED)
    with pytest.raises(api.SqgError) as e:
        gen.sample(2)
    assert e.value.code == -1 and "sqg_genome_load" in str(e.value)
    with pytest.raises(api.SqgError):
        gen.load_genome([b"ACGT" * 100], 0)                       # -r must be positive
    gen.load_genome([b"ACGT" * 100], 500)                          # 400 nt: reads get clipped, still >= 200
    bs = [gen.sample(2).run().wait() for _ in range(3)]
    enc, off = bs[2].compress()
    assert off[-1] == len(enc) > 0
    with pytest.raises(api.SqgError) as e:                         # batch 0's slot was reused by batch 2
        bs[0].compress()
    assert e.value.code == -4
    with pytest.raises(api.SqgError) as e:
        gen.sample(2, workers=np.array([0, 5], np.int32))          # worker 5 does not exist
    assert e.value.code == -1
    for b in bs:
        b.free()
    gen.close()


MFREQ = os.path.join(INPUTS, "mfreq.tsv")
MFREQ_DENSE = os.path.join(INPUTS, "mfreq_dense.tsv")


@pytest.mark.gpu
@pytest.mark.parametrize("T,batches,rlen,mf", [(1, [6, 5], 4000, MFREQ), (1, [40, 30], 1500, MFREQ_DENSE), (8, [8, 8, 8], 2500, MFREQ_DENSE),
                                              (3, [20, 33], 1200, MFREQ_DENSE)], ids=["t1_ref_file", "t1_long_chain", "tk8", "t3"])
def test_cpg_methylation_in_the_device_sampler(T, batches, rlen, mf):
    """--meth-freq: every CpG of a read's reference span draws from the worker's rand_meth stream in order; methylated Cs
    become 'M' (also on the '-' strand), the signal comes from the 5-letter table (src/genread.c:207-241, src/seq.h:45-74)"""
    _run("dna-r9-prom", 6, NCOV, T, batches, rlen=rlen, meth_freq=mf)


@pytest.mark.gpu
def test_methylation_only_draws_for_contigs_with_a_frequency_array(tmp_path):
    """two contigs, frequencies for one of them only: reads of the other take no rand_meth draws (src/genread.c:208)"""
    rng = np.random.default_rng(4)
    contigs = [bytes(rng.choice(list(b"ACGT"), n).astype(np.uint8)) for n in (9000, 7000)]
    fa = tmp_path / "g.fa"
    fa.write_text("".join(f">c{i}\n{c.decode()}\n" for i, c in enumerate(contigs)))
    cpg = [i for i in range(len(contigs[1]) - 1) if contigs[1][i:i + 2] == b"CG"]
    mf = tmp_path / "m.tsv"
    mf.write_text("".join(f"c1\t{p}\t{(j % 11) / 10:.1f}\n" for j, p in enumerate(cpg[::2])))
    _run("dna-r9-prom", 6, str(fa), 2, [10, 10, 7], rlen=1200, seed=9, meth_freq=str(mf))
