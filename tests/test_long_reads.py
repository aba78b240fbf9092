"""Chromosome-scale reads (`--full-contigs` on a real genome: one read per contig, src/genread.c:311-355 / src/sim.c:520-540) and
the reference's length guard (src/sim.c:559-562: a read of >= UINT32_MAX samples is an error).  A read far longer than a link
is cut into pieces of whole 512-event segments that many wavefronts walk concurrently (k_part_events.h); the pieces' tile
offsets are rebased by k_part_tile_bases and the read's totals are 64-bit sums of the pieces'."""
import numpy as np
import pytest

import orc
from squigulator_amd import api, model, profiles


def _seq(rng, n):
    return bytes(rng.choice(list(b"ACGT"), int(n)).astype(np.uint8))


@pytest.mark.gpu
@pytest.mark.parametrize("name,T,mb", [("dna-r10-prom", 1, 5.3), ("dna-r9-prom", 1, 5.1), ("dna-r10-prom", 2, 2.2)])
def test_chromosome_scale_read_matches_the_oracle(name, T, mb):
    rng = np.random.default_rng(int(mb * 10))
    prof, fl = profiles.get_profile(name)
    k = profiles.default_kmer_size(fl)
    mean, stdv = model.synthetic_model(k)
    # a few ordinary reads, the long one in the middle of the worker's chain, ordinary reads behind it (they continue its streams)
    reads = [_seq(rng, m) for m in rng.integers(300, 4000, 12)] + [_seq(rng, int(mb * 1e6))] + [_seq(rng, m) for m in rng.integers(300, 4000, 12)]
    batches = [reads, [_seq(rng, m) for m in rng.integers(300, 4000, 40)]]
    orac = orc.Oracle(prof, fl, k, mean, stdv, 42, num_workers=T)
    want = [orac.run_batch_seqs(bt) for bt in batches]
    orac.close()
    gen = api.SignalGenerator(prof, fl, k, mean, stdv, 42, num_workers=T, mode=api.MODE_CERTIFIED)
    for bi, bt in enumerate(batches):
        b = gen.submit(bt)
        sig, dw = b.signal(), b.dwell()
        assert b.n_samples == sum(len(w.sig) for w in want[bi])
        for i, w in enumerate(want[bi]):
            np.testing.assert_array_equal(sig[b.sig_off[i]:b.sig_off[i + 1]], w.sig, err_msg=f"batch {bi} read {i} ({len(bt[i])} nt)")
            np.testing.assert_array_equal(dw[b.ev_off[i]:b.ev_off[i + 1]], w.ss)
        b.free()
    gen.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["dna-r9-prom", "dna-r10-prom", "rna-r9-prom"])
def test_a_read_of_uint32_max_samples_is_an_error_not_a_crash(name):
    """src/sim.c:559-562: `len_raw_signal >= UINT32_MAX` ends the reference's run; here the batch fails with SQG_EOVERFLOW (the
    sample kernels leave such a read alone: its 32-bit positions would wrap) and the context goes on with the next batch"""
    rng = np.random.default_rng(3)
    prof, fl = profiles.get_profile(name)
    k = profiles.default_kmer_size(fl)
    mean, stdv = model.synthetic_model(k)
    # --ideal-time (src/gensig.c:254: sps = dwell_mean for every event) with a dwell of 50000 samples: 86000 events are enough
    long_prof = prof.replace(dwell_mean=50000.0, dwell_std=0.0)
    flags = fl | profiles.SQ_IDEAL_TIME
    gen = api.SignalGenerator(long_prof, flags, k, mean, stdv, 42, num_workers=1, mode=api.MODE_CERTIFIED)
    ok = [_seq(rng, 40), _seq(rng, 25)]
    b = gen.submit(ok)                                              # 2e6 samples per short read: fine
    assert b.n_samples == 50000 * sum(max(len(r) - k + 1, 5) for r in ok)
    b.free()
    with pytest.raises(api.SqgError) as ei:
        gen.submit([_seq(rng, 30), _seq(rng, 86000 + k), _seq(rng, 30)])
    assert ei.value.code == -6, str(ei.value)                       # SQG_EOVERFLOW
    b = gen.submit(ok)                                              # the context is still usable
    assert b.n_samples == 50000 * sum(max(len(r) - k + 1, 5) for r in ok)
    b.free()
    gen.close()
