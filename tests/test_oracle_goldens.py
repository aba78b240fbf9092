"""Pin 1 of the oracle: the reference's OWN regression goldens (test/*.exp, the files
scripts/test.sh:24-139 diffs), reproduced by the oracle through the product's text writers.

The built-in ONT pore-model tables are absent from the mounted reference, so the raw_signal
column (amplitudes) is masked; every other byte -- header, read ids (contig/start/end/strand =
the read sampler), offset, len_raw_signal (= all dwell draws), median_before, read_number,
start_time, FASTA, PAF incl. the full per-event dwell string, SAM -- must match exactly.
That pins: the LCG, Box-Muller, Erlang draw, seeding layout, sampler incl. N/short rejection,
dwell draw/fold, prefix/stall event counts, RNA bookkeeping and the text formatting.
Amplitude arithmetic is pinned by test_oracle_vs_ref.py and test_golden_amplitude_consistency.py.
"""
import gzip
import os

import pytest

import simrun
from refcases import CASES

EXP = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_exp")


def _exp(name):
    with gzip.open(os.path.join(EXP, name + ".gz"), "rt") as f:
        return f.read()


@pytest.mark.parametrize("cid,cmd,exp", CASES, ids=[c[0] for c in CASES])
def test_reference_goldens(cid, cmd, exp):
    o, k, names, lengths, reads, orac = simrun.run_oracle(cmd)
    got = simrun.format_outputs(o, k, names, lengths, reads, mask_signal=True)
    orac.close()
    for kind, fname in exp.items():
        want = _exp(fname)
        if kind == "slow5":
            want = simrun.mask_slow5_signal(want)
        assert got[kind] == want, f"{cid}: {kind} differs from {fname}"
