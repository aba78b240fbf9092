"""Pin 2 of the oracle: sample-for-sample equality with the REFERENCE's own gensig.c/genread.c.

The vectors in tests/golden/refvec/*.npz were produced in the build container by
oracle/_ref/ref_harness (the reference's translation units compiled where they lie, driven by
oracle/ref_harness.c) -- see tools/make_refvec.py.  When the harness binary and /root/reference
are present the comparison is additionally made live, including a randomized case.
"""
import os
import sys
import tempfile

import numpy as np
import pytest

import orc
import simrun
from refvec_cases import REFVEC_CASES
from squigulator_amd import model, options

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
VEC = os.path.join(HERE, "golden", "refvec")
HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")


def _check(reads, v):
    meta = v["meta"]
    assert len(reads) == len(meta)
    so = go = eo = 0
    for i, r in enumerate(reads):
        tid, ref_idx, ref_len, pos_st, rlen, strand, start_time, nsig, nss = (int(x) for x in meta[i])
        assert (r.tid, r.ref_idx, r.ref_pos_st, r.rlen, ord(r.strand)) == (tid, ref_idx, pos_st, rlen, strand), i
        assert r.seq == v["seq"][so:so + rlen].tobytes(), f"read {i}: sequence differs"
        assert r.offset == v["offset"][i] and r.median_before == v["median"][i], f"read {i}: offset/median"
        assert r.start_time == start_time
        assert len(r.sig) == nsig, f"read {i}: len_raw_signal {len(r.sig)} != {nsig}"
        np.testing.assert_array_equal(r.sig, v["sig"][go:go + nsig], err_msg=f"read {i}: raw_signal")
        np.testing.assert_array_equal(r.ss, v["ss"][eo:eo + nss], err_msg=f"read {i}: per-event dwell")
        so += rlen; go += nsig; eo += nss


@pytest.mark.parametrize("cid,cmd", REFVEC_CASES, ids=[c[0] for c in REFVEC_CASES])
def test_oracle_matches_committed_reference_vectors(cid, cmd):
    v = np.load(os.path.join(VEC, cid + ".npz"))
    assert str(v["cmd"]) == cmd, "fixture was generated from a different command line; rerun tools/make_refvec.py"
    o, k, names, lengths, reads, orac = simrun.run_oracle(cmd)
    assert k == int(v["k"])
    _check(reads, v)
    orac.close()


def test_oracle_thread_count_invariance():
    """host threads over virtual workers must not change anything (T=K regime)."""
    cmd = "nCoV-2019.reference.fasta -x dna-r9-prom -n 40 --seed 42 -r 600 -t 16 -K 16"
    v = np.load(os.path.join(VEC, "r9_tk16.npz"))
    o, k, names, lengths, reads, orac = simrun.run_oracle(cmd, nthreads=4)
    _check(reads, v)
    orac.close()


@pytest.mark.skipif(not (os.path.exists(HARNESS) and os.path.isdir("/root/reference")),
                    reason="compiled reference harness only exists in the build container")
@pytest.mark.parametrize("seed", [3, 12345])
def test_oracle_matches_live_reference(seed):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_refvec
    cmd = f"nCoV-2019.reference.fasta -x dna-r10-prom -n 6 --seed {seed} -r 800 -t 3 -K 3 --amp-noise 1.7 --dwell-std 6"
    with tempfile.TemporaryDirectory() as tmp:
        make_refvec._model_files.clear()
        ref_reads, k = make_refvec.run_harness(cmd, tmp)
    v = make_refvec.pack(ref_reads)
    o, k2, names, lengths, reads, orac = simrun.run_oracle(cmd)
    _check(reads, v)
    orac.close()
