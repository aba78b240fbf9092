"""Parity at BASELINE.json's full batch size through size-independent properties:
  * the certified fp32 path and the FP64 path give the same 5.6e8-sample stream (bit for bit);
  * two independent contexts give the same stream (determinism, no cross-workgroup race);
  * per-read invariants (dwell sums, offsets) hold for all 8192 reads;
  * a random subset of reads equals the oracle (each read's worker is independent in the T=K regime);
  * a second batch on the same context (stream states carried across batches) still matches the oracle."""
import os
import sys

import numpy as np
import pytest

import orc
from squigulator_amd import api, model, profiles

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

K = 8192


@pytest.fixture(scope="module")
def workload():
    import bench
    genome = bench.load_genome(bench.GENOME)
    rng = np.random.default_rng(2024)
    return [bench.sample_reads(genome, K, 10000, rng) for _ in range(2)]


def _run(workload, mode, n_batches=1):
    prof, fl = profiles.get_profile("dna-r9-prom")
    mean, stdv = model.synthetic_model(6)
    gen = api.SignalGenerator(prof, fl, 6, mean, stdv, seed=42, num_workers=K, mode=mode)
    outs = []
    for reads in workload[:n_batches]:
        b = gen.submit(reads)
        outs.append((b.signal(), b.sig_off.copy(), b.dwell(), b.ev_off.copy(), b.offset.copy(), b.median_before.copy()))
        b.free()
    fb = gen.timing()["fallback_samples"]
    gen.close()
    return outs, fb


def test_full_batch_properties(workload):
    (cert, fb) = _run(workload, api.MODE_CERTIFIED, 2)
    (exact, _) = _run(workload, api.MODE_EXACT, 2)
    (again, _) = _run(workload, api.MODE_CERTIFIED, 1)
    prof, fl = profiles.get_profile("dna-r9-prom")
    mean, stdv = model.synthetic_model(6)
    for bi in range(2):
        sig, off, dw, eoff, offs, med = cert[bi]
        assert len(sig) > 4e8 and off[-1] == len(sig)
        # certified == exact, everywhere
        assert np.array_equal(sig, exact[bi][0]) and np.array_equal(dw, exact[bi][2]) and np.array_equal(off, exact[bi][1])
        # per-read: samples == sum of per-event dwell, events == len - k + 1
        ev_sum = np.add.reduceat(dw.astype(np.int64), eoff[:-1])
        assert np.array_equal(ev_sum, np.diff(off))
        assert np.array_equal(np.diff(eoff), np.array([len(r) - 6 + 1 for r in workload[bi]]))
        assert dw.min() >= 1
    assert np.array_equal(cert[0][0], again[0][0]), "two contexts, same seed, same batch: streams differ"
    assert 0 < fb < 1e-3 * len(cert[1][0])

    # oracle on a random subset of workers (both batches: the second needs the first's stream states)
    rng = np.random.default_rng(7)
    pick = sorted(rng.choice(K, size=24, replace=False).tolist())
    o = orc.Oracle(prof, fl, 6, mean, stdv, 42, num_workers=K)
    for bi in range(2):
        sig, off, dw, eoff, offs, med = cert[bi]
        res = o.run_batch_assigned([workload[bi][i] for i in pick], pick, want_ss=True)
        for i, r in zip(pick, res):
            assert np.array_equal(sig[off[i]:off[i + 1]], r.sig), f"batch {bi} read {i}"
            assert np.array_equal(dw[eoff[i]:eoff[i + 1]], r.ss)
            assert offs[i] == r.offset and med[i] == r.median_before
    o.close()


@pytest.mark.parametrize("name,extra_flags,kk,n", [
    ("dna-r10-prom", 0, 9, 1024),
    ("rna004-prom", profiles.SQ_PREFIX, 9, 1536),
    ("rna-r9-prom", profiles.SQ_PREFIX, 5, 1024),
], ids=["r10_9mer", "rna004_prefix", "rna_r9_prefix"])
def test_medium_batches_other_chemistries(name, extra_flags, kk, n):
    """9-mer (hashed bins, stream states in HBM) and RNA (+prefix: level-shift window, stall segment, reversal,
    dwell up to ~270 samples/event) at a few 10^8 samples: certified == exact, subset == oracle, two batches."""
    import bench
    prof, fl = profiles.get_profile(name)
    fl |= extra_flags
    mean, stdv = model.synthetic_model(kk)
    rng = np.random.default_rng(99)
    if fl & profiles.SQ_RNA:
        seqs = []
        with open(os.path.join(ROOT, "tests", "golden", "inputs", "rnasequin_sequences_2.4.fa")) as f:
            cur = []
            for line in f:
                if line.startswith(">"):
                    if cur:
                        seqs.append("".join(cur).encode())
                    cur = []
                else:
                    cur.append(line.strip())
            seqs.append("".join(cur).encode())
        batches = [[seqs[int(i)] for i in rng.integers(0, len(seqs), size=n)] for _ in range(2)]
    else:
        genome = bench.load_genome(bench.GENOME)
        batches = [bench.sample_reads(genome, n, 10000, rng) for _ in range(2)]
    outs = {}
    for mode in (api.MODE_CERTIFIED, api.MODE_EXACT):
        gen = api.SignalGenerator(prof, fl, kk, mean, stdv, seed=5, num_workers=n, mode=mode)
        outs[mode] = []
        for reads in batches:
            b = gen.submit(reads)
            outs[mode].append((b.signal(), b.sig_off.copy(), b.dwell(), b.ev_off.copy(), b.offset.copy()))
            b.free()
        gen.close()
    pick = sorted(rng.choice(n, size=12, replace=False).tolist())
    o = orc.Oracle(prof, fl, kk, mean, stdv, 5, num_workers=n)
    for bi in range(2):
        sig, off, dw, eoff, offs = outs[api.MODE_CERTIFIED][bi]
        assert np.array_equal(sig, outs[api.MODE_EXACT][bi][0]) and np.array_equal(dw, outs[api.MODE_EXACT][bi][2])
        res = o.run_batch_assigned([batches[bi][i] for i in pick], pick, want_ss=True)
        for i, r in zip(pick, res):
            assert np.array_equal(sig[off[i]:off[i + 1]], r.sig), f"{name} batch {bi} read {i}"
            assert np.array_equal(dw[eoff[i]:eoff[i + 1]], r.ss) and offs[i] == r.offset
    o.close()
