"""BASELINE.json configs[2]: `-x dna-r10-prom` (R10 9-mer table) on the hg38-proportioned multi-contig genome with N runs,
reads drawn by the device-side gen_read (src/genread.c:179-194,243-281), in the regime bench.py runs (`-t 1`: one worker
chain cut into links, 9-mer streams handed out over bucketed events, squigulator_amd/csrc/k_part.h) and with one worker per read.

* a scaled-down genome of the same layout (bench.synthetic_genome_host): sampler coordinates, sequences, dwells and signals
  against the oracle, two batches (carried stream state);
* certified == exact over whole batches;
* one run on the full-size genome (24 contigs with hg38's lengths, 3.09 Gb, made in HBM): coordinates against the genome,
  certified == exact, and the whole first batch against the oracle."""
import os
import sys

import numpy as np
import pytest

import orc
from squigulator_amd import api, model, profiles

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from test_sampler import _run  # noqa: E402


def _fasta(tmp_path, mb):
    contigs = bench.synthetic_genome_host(mb)
    fa = tmp_path / "g.fa"
    with open(fa, "wb") as f:
        for i, c in enumerate(contigs):
            f.write(b">chr%d\n" % (i + 1) + c + b"\n")
    return str(fa), contigs


@pytest.mark.gpu
@pytest.mark.parametrize("T,batches,rlen", [(1, [40, 30], 3000), (1, [300], 900), (24, [24, 24], 3000), (3, [36, 24], 2500)],
                         ids=["t1", "t1_many_short", "tk24", "t3"])
def test_r10_on_the_hg38_layout_matches_the_oracle(T, batches, rlen, tmp_path):
    fa, contigs = _fasta(tmp_path, 3.0)
    assert sum(c.count(b"N") for c in contigs) > 0.03 * sum(len(c) for c in contigs)
    n = _run("dna-r10-prom", 9, fa, T, batches, rlen=rlen)
    assert n >= 24


@pytest.mark.gpu
@pytest.mark.parametrize("T", [1, 512], ids=["t1", "tk"])
def test_certified_equals_exact_over_whole_batches(T, tmp_path):
    _, contigs = _fasta(tmp_path, 8.0)
    prof, fl = profiles.get_profile("dna-r10-prom")
    mean, stdv = model.synthetic_model(9)
    out = []
    for mode in (api.MODE_CERTIFIED, api.MODE_EXACT):
        gen = api.SignalGenerator(prof, fl, 9, mean, stdv, 42, num_workers=T, mode=mode)
        gen.load_genome(contigs, 6000, api.SAMPLE_DNA)
        res = []
        for _ in range(2):
            b = gen.sample(512).run().wait()
            res.append((b.signal().copy(), np.array(b.sig_off), b.dwell().copy(), dict(b.sampled)))
            b.free()
        gen.close()
        out.append(res)
    for (s0, o0, d0, m0), (s1, o1, d1, m1) in zip(*out):
        np.testing.assert_array_equal(o0, o1)
        np.testing.assert_array_equal(d0, d1)
        np.testing.assert_array_equal(m0["ref_pos"], m1["ref_pos"])
        np.testing.assert_array_equal(s0, s1)
        assert len(s0) > 1.0e7


# the two full-size checks (3.09 Gb resident: tests/fullsize_hg38.py against the oracle, tests/batchsize_hg38.py across batch sizes) are
# tests/test_00_configs.py::test_config2_hg38_r10_* -- collected first, named after the config
