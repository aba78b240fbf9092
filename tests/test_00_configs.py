"""One test per BASELINE.json config, named after it, collected FIRST (tests/conftest.py orders this file ahead of everything else):
the driver's `pytest -x -q -m gpu` record says, from its first lines alone, whether the HIP path reproduces the reference on every
workload `configs` names.  Each config is checked twice:

  * against the oracle (oracle/sqg_oracle.c, pinned by the reference's goldens and by the compiled reference's vectors) or against the
    compiled reference's committed vectors (tests/golden/refvec) at a size the oracle finishes in seconds -- coordinates, sequences,
    per-event dwells, offsets, medians and every int16, bit for bit, through the C ABI;
  * at the config's full size through a size-independent property: the two arithmetic paths of the library (MODE_EXACT: all FP64;
    MODE_CERTIFIED: fp32 with an acceptance test and FP64 fix-ups) give the same digest over the whole job, a read's samples are the sum
    of its dwells, `-t 1` does not know the batch size, a rank's shard is the same reads as the whole job's.

No clocks, no rates: deterministic comparisons only (the reference's bar: scripts/test.sh:24-139)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import xxhash

import hiprun
import orc
from refvec_cases import REFVEC_CASES
from squigulator_amd import api, model, profiles
from test_hip_parity import VEC, _compare, _fixture_reads
from test_sampler import NCOV, SEQUIN, _contigs

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
NCPU = max(1, min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 32))


def _against_oracle(profile, k, fasta, T, batches, rlen, sflags=0, mode=api.SAMPLE_DNA, gen_mode=api.MODE_CERTIFIED, seed=42):
    """the device sampler + the generator against the oracle's gen_read + gen_sig, batch by batch (carried stream state); returns
    (reads, samples) compared"""
    prof, fl = profiles.get_profile(profile)
    mean, stdv = model.synthetic_model(k)
    orac = orc.Oracle(prof, fl | sflags, k, mean, stdv, seed, num_workers=T, rlen=rlen)
    ref = orac.load_ref(fasta, None, None)
    gen = api.SignalGenerator(prof, fl | sflags, k, mean, stdv, seed, num_workers=T, mode=gen_mode)
    gen.load_genome(_contigs(ref), rlen, mode, None)
    n_reads = n_samples = 0
    for nb in batches:
        want = orac.run_batch(nb, want_ss=True, nthreads=NCPU if T > 1 else 1)
        b = gen.sample(nb).run().wait()
        s, seqs, sig, dw = b.sampled, b.reads(), b.signal(), b.dwell()
        so, eo = np.array(b.sig_off), np.array(b.ev_off)
        assert len(want) == nb == len(seqs)
        for i, w in enumerate(want):
            assert (s["ref_idx"][i], s["ref_pos"][i], s["rlen"][i], chr(s["strand"][i])) == (w.ref_idx, w.ref_pos_st, w.rlen, w.strand), f"read {i}"
            assert seqs[i] == w.seq, f"read {i}: sequence"
            assert so[i + 1] - so[i] == len(w.sig), f"read {i}: len_raw_signal"
            assert np.array_equal(dw[eo[i]:eo[i + 1]], w.ss), f"read {i}: per-event dwell"
            assert np.array_equal(sig[so[i]:so[i + 1]], w.sig), f"read {i}: raw_signal"
            assert b.offset[i] == w.offset and b.median_before[i] == w.median_before, f"read {i}: offset / median_before"
        n_reads += nb
        n_samples += int(so[-1])
        b.free()
    gen.close(); orac.close()
    return n_reads, n_samples


def _job_digest(profile, k, contigs, T, n, K, rlen, gen_mode, sflags=0, mode=api.SAMPLE_DNA, seed=42, device_digest=True):
    """a whole job (n reads in batches of K) through the C ABI: a digest of every batch's signal, offsets, dwells and coordinates; the
    sum-of-dwells property on every read"""
    prof, fl = profiles.get_profile(profile)
    mean, stdv = model.synthetic_model(k)
    gen = api.SignalGenerator(prof, fl | sflags, k, mean, stdv, seed, num_workers=T, mode=gen_mode)
    gen.load_genome(contigs, rlen, mode, None)
    h = xxhash.xxh3_128()                                        # (14 GB of int16 per mode at configs[1]'s size: sha256 would be most of the test's time)
    done = samples = 0
    while done < n:
        nb = min(K, n - done)
        b = gen.sample(nb).run().wait()
        so, eo, dw = np.array(b.sig_off), np.array(b.ev_off), b.dwell()
        per_read = np.add.reduceat(dw.astype(np.int64), eo[:-1]) if len(dw) else np.zeros(nb, np.int64)
        assert np.array_equal(per_read, np.diff(so)), "a read's samples are the sum of its dwells (src/gensig.c:254-272)"
        h.update(b.signal().tobytes()); h.update(so.tobytes()); h.update(dw.tobytes())
        h.update(np.array(b.offset).tobytes()); h.update(np.array(b.median_before).tobytes())
        for key in ("ref_idx", "ref_pos", "rlen"):
            h.update(np.array(b.sampled[key]).tobytes())
        samples += int(so[-1])
        done += nb
        b.free()
    gen.close()
    return h.hexdigest(), samples


def _ncov_contigs():
    prof, fl = profiles.get_profile("dna-r9-prom")
    mean, stdv = model.synthetic_model(6)
    o = orc.Oracle(prof, fl, 6, mean, stdv, 42, num_workers=1, rlen=10000)
    c = _contigs(o.load_ref(NCOV, None, None))
    o.close()
    return c


def _sequin_contigs():
    prof, fl = profiles.get_profile("rna004-prom")
    mean, stdv = model.synthetic_model(9)
    o = orc.Oracle(prof, fl, 9, mean, stdv, 42, num_workers=1, rlen=10000)
    c = _contigs(o.load_ref(SEQUIN, None, None))
    o.close()
    return c


# ---- configs[0]: nCoV-2019.reference.fasta -x dna-r9-prom -n 1000 --seed 42 (the reference's CPU-runnable case) ----------------------

@pytest.mark.parametrize("mode", [api.MODE_EXACT, api.MODE_CERTIFIED], ids=["exact", "certified"])
def test_config0_ncov_r9_n1000_seed42_reference_vectors(mode):
    """the compiled reference's own output (oracle/_ref, tools/make_refvec.py -> tests/golden/refvec/r9_t1.npz and r9_tk16.npz): the HIP
    path on the same reads, every int16"""
    for cid in ("r9_t1", "r9_tk16"):
        cmd = dict(REFVEC_CASES)[cid]
        want = _fixture_reads(np.load(os.path.join(VEC, cid + ".npz")))
        got = hiprun.run_hip_on_reads(cmd, [w["seq"] for w in want], mode=mode)
        _compare(got, want, cid)


def test_config0_ncov_r9_n1000_seed42_whole_job_against_the_oracle():
    """the config's own command line at its own size, `-t 1 -K 1000 -r 10000 --seed 42`: 1000 reads, ~7e7 samples, read by read"""
    n, ns = _against_oracle("dna-r9-prom", 6, NCOV, 1, [1000], rlen=10000)
    assert n == 1000 and ns > 5.0e7


# ---- configs[1]: nCoV-2019 -x dna-r9-prom -n 100000 on 1 x MI355X (R9 6-mer model in LDS) ---------------------------------------------

def test_config1_ncov_r9_n100000_tk8192_batch_against_the_oracle():
    """the regime bench.py --workload ncov-r9 runs (T = K: one virtual worker per read, its 4096 streams in LDS): one 8192-read batch and
    the batch behind it (every worker's carried state), 1.1e9 samples, read by read"""
    n, ns = _against_oracle("dna-r9-prom", 6, NCOV, 8192, [8192, 4096], rlen=10000)
    assert n == 12288 and ns > 7.0e8


def test_config1_ncov_r9_n100000_full_size_certified_equals_exact():
    """-n 100000 in full (T = K = 8192, 13 batches, ~7e9 samples): the digest of the all-FP64 path == the digest of the certified path"""
    c = _ncov_contigs()
    a, na = _job_digest("dna-r9-prom", 6, c, 8192, 100000, 8192, 10000, api.MODE_CERTIFIED)
    b, nb = _job_digest("dna-r9-prom", 6, c, 8192, 100000, 8192, 10000, api.MODE_EXACT)
    assert na == nb and na > 6.0e9 and a == b


# ---- configs[2]: hg38noAlt.fa -x dna-r10-prom -n 1000000 on 1 x MI355X (R10 9-mer model) ---------------------------------------------

def _script(name, ok):
    p = subprocess.run([sys.executable, os.path.join(HERE, name)], capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    assert ok in p.stdout, p.stdout[-2000:]


def test_config2_hg38_r10_full_size_genome_t1_against_the_oracle():
    """3.09 Gb resident (24 contigs with hg38's lengths), `-t 1 -K 1024` x 2: coordinates against the genome, certified == exact, the whole
    first batch against the oracle (tests/fullsize_hg38.py, a process of its own)"""
    _script("fullsize_hg38.py", "full-size ok")


def test_config2_hg38_r10_headline_batch_equals_small_batches():
    """the bench's 32768-read batch == 2 x 16384 == (first 2048 reads) 2 x 1024, the size compared with the oracle above: 4.3e9 int16
    (tests/batchsize_hg38.py)"""
    _script("batchsize_hg38.py", "batch-size ok")


# ---- configs[3]: hg38noAlt.fa -x dna-r10-prom -n 9000000 sharded across 8 x MI355X ----------------------------------------------------

def test_config3_hg38_r10_8gpu_shard_of_rank_r_against_the_oracle(tmp_path):
    """`-t 8`: GPU g runs virtual worker g (squigulator_amd/shard.py, the reference's static partition src/thread.c:84-101: reads
    [g K/8, (g+1) K/8) of every batch).  Rank g's context -- ONE worker seeded as worker g of 8 -- on its slice of two batches against
    the oracle's `-t 8` run of the whole batches, for g = 0, 3 and 7, on the hg38-proportioned genome (scaled: the oracle draws the reads)"""
    import bench
    from squigulator_amd import shard
    contigs = bench.synthetic_genome_host(3.0)
    fa = tmp_path / "g.fa"
    with open(fa, "wb") as f:
        for i, c in enumerate(contigs):
            f.write(b">chr%d\n" % (i + 1) + c + b"\n")
    prof, fl = profiles.get_profile("dna-r10-prom")
    mean, stdv = model.synthetic_model(9)
    G, K, rlen = 8, 256, 3000
    orac = orc.Oracle(prof, fl, 9, mean, stdv, 42, num_workers=G, rlen=rlen)
    orac.load_ref(str(fa), None, None)
    want = [orac.run_batch(K, want_ss=True, nthreads=NCPU) for _ in range(2)]
    orac.close()
    step = K // G
    n_cmp = 0
    for g in (0, 3, 7):
        lo, hi = shard.worker_range(g, G, G)
        assert (lo, hi) == (g, g + 1)
        gen = api.SignalGenerator(prof, fl, 9, mean, stdv, 42, num_workers=G, mode=api.MODE_CERTIFIED, worker_lo=lo, worker_hi=hi)
        for wb in want:
            mine = wb[g * step:(g + 1) * step]
            assert all(r.tid == g for r in mine)
            b = gen.stage([r.seq for r in mine], np.full(len(mine), g, np.int32)).run().wait()
            sig, dw = b.signal(), b.dwell()
            for i, w in enumerate(mine):
                assert np.array_equal(sig[b.sig_off[i]:b.sig_off[i + 1]], w.sig), f"rank {g} read {i}"
                assert np.array_equal(dw[b.ev_off[i]:b.ev_off[i + 1]], w.ss)
                assert b.offset[i] == w.offset and b.median_before[i] == w.median_before
            n_cmp += int(b.n_samples)
            b.free()
        gen.close()
    assert n_cmp > 3 * 2 * step * 30000


# ---- configs[4]: rnasequin_sequences_2.4.fa -x rna004-prom -n 1000000 on 8 x MI355X (whole transcripts, polyA/adaptor) ----------------

def test_config4_sequin_rna004_prefix_against_the_oracle():
    """`--prefix=yes`: polyA + adaptor attached, the adaptor's level shift, the stall, the reversed store (src/genread.c:71-123,
    src/gensig.c:349-353); `-t 1` (one worker per GPU, bench.py --workload sequin-rna004) two chained batches, and T = K"""
    n, ns = _against_oracle("rna004-prom", 9, SEQUIN, 1, [600, 424], rlen=10000, sflags=profiles.SQ_PREFIX, mode=api.SAMPLE_RNA)
    assert n == 1024 and ns > 3.0e7
    n, ns = _against_oracle("rna004-prom", 9, SEQUIN, 256, [256, 256], rlen=10000, sflags=profiles.SQ_PREFIX, mode=api.SAMPLE_RNA)
    assert n == 512 and ns > 1.5e7


def test_config4_sequin_rna004_full_batch_certified_equals_exact():
    """a 32768-read batch (the bench's step) and its successor, `-t 1 --prefix=yes`: all-FP64 digest == certified digest, ~3e9 samples"""
    c = _sequin_contigs()
    a, na = _job_digest("rna004-prom", 9, c, 1, 65536, 32768, 10000, api.MODE_CERTIFIED, sflags=profiles.SQ_PREFIX, mode=api.SAMPLE_RNA)
    b, nb = _job_digest("rna004-prom", 9, c, 1, 65536, 32768, 10000, api.MODE_EXACT, sflags=profiles.SQ_PREFIX, mode=api.SAMPLE_RNA)
    assert na == nb and na > 2.0e9 and a == b
