"""Batches of many short reads: the read-offset scan (k_scan) runs over many workgroups that pass their totals on."""
import numpy as np
import pytest

import orc
from squigulator_amd import api, model, profiles


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 1023, 1024, 1025, 5000])
def test_offsets_of_many_short_reads(n):
    rng = np.random.default_rng(n)
    prof, fl = profiles.get_profile("dna-r9-prom")
    mean, stdv = model.synthetic_model(6)
    reads = [bytes(rng.choice(list(b"ACGT"), int(m)).astype(np.uint8)) for m in rng.integers(6, 40, n)]
    gen = api.SignalGenerator(prof, fl, 6, mean, stdv, 9, num_workers=n, mode=api.MODE_CERTIFIED)
    for rep in range(2):                                     # twice: the tickets of the first launch are still in memory
        b = gen.submit(reads)
        sig, dw = b.signal(), b.dwell()
        lens = np.array([dw[b.ev_off[i]:b.ev_off[i + 1]].sum() for i in range(n)])
        np.testing.assert_array_equal(np.diff(b.sig_off), lens)
        assert b.sig_off[0] == 0 and b.sig_off[-1] == len(sig) == b.n_samples
        if rep == 0 and n <= 1025:
            o = orc.Oracle(prof, fl, 6, mean, stdv, 9, num_workers=n)
            want = o.run_batch_seqs(reads)
            o.close()
            for i in (0, n // 2, n - 1):
                np.testing.assert_array_equal(sig[b.sig_off[i]:b.sig_off[i + 1]], want[i].sig)
        b.free()
    gen.close()
