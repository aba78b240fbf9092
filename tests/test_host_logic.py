"""Host-side mirrors of the reference's option / profile / model / text-format logic."""
import numpy as np
import pytest

from squigulator_amd import model, options, profiles, shard, slow5_text


def test_profile_presets_match_reference_literals():
    p, fl = profiles.get_profile("dna-r10-prom")                      # src/sim.c:102-114
    assert (p.digitisation, p.sample_rate, p.range, p.dwell_mean, p.dwell_std) == (2048, 5000, 281.345551, 13.0, 4.0)
    assert fl == profiles.SQ_R10
    p, fl = profiles.get_profile("rna004-prom")                       # src/sim.c:126-137
    assert p.dwell_std == 0.0 and fl == profiles.SQ_R10 | profiles.SQ_RNA
    assert profiles.default_kmer_size(0) == 6 and profiles.default_kmer_size(profiles.SQ_RNA) == 5
    assert profiles.default_kmer_size(profiles.SQ_R10) == 9
    with pytest.raises(ValueError):
        profiles.get_profile("nope")


def test_option_reconciliation():
    o = options.parse_args("ref.fa -x dna-r10-prom --bps 200 -n 2 --seed 1 -t1")   # src/sim.c:1040-1045
    assert o.profile.dwell_mean == 25.0 and o.threads == 1 and o.nreads == 2
    o = options.parse_args("ref.fa -x rna004-min --dwell-mean 30 --dwell-std 3.0")  # src/sim.c:1035-1038
    assert o.profile.bps == 133.0 and o.profile.dwell_mean == 30.0
    # a bare --trans-trunc swallows the next token, as getopt's required_argument does (scripts/test.sh:117)
    o = options.parse_args("-x rna004-prom -n 1 --seed 1 --trans-trunc -t1 ref.fa")
    assert o.threads == 8 and not (o.flags & profiles.SQ_TRANS_TRUNC)
    o = options.parse_args("ref.fa -r 50")
    assert o.rlen == 200                                              # src/sim.c:930-933
    assert options.resolve_nreads(options.parse_args("ref.fa -f 2 -r 20000"), 1, 29903) == 2


def test_double_to_str():
    f = slow5_text.double_to_str                                      # slow5_misc.c:379-405
    assert f(2048.0) == "2048" and f(-272.365141) == "-272.365141" and f(748.5801) == "748.5801"
    assert f(-0.0000001) == "0" and f(0.5) == "0.5" and f(100.0) == "100"


def test_model_roundtrip(tmp_path):
    for k in (5, 6):
        mean, stdv = model.synthetic_model(k)
        assert mean.dtype == np.float32 and len(mean) == 4 ** k
        assert mean.min() >= 60 and mean.max() < 140 and stdv.min() >= 1 and stdv.max() < 4
        p = tmp_path / f"m{k}.model"
        model.write_f5c_model(p, k, mean, stdv)
        k2, m2, s2 = model.read_f5c_model(p)
        assert k2 == k and (m2 == mean).all() and (s2 == stdv).all()
    assert model.kmer_string(0b000110, 3) == "ACG"
    bad = tmp_path / "bad.model"
    bad.write_text("kmer\tlevel_mean\tlevel_stdv\nAAAAA\t1.0\t2.0\n")
    with pytest.raises(ValueError):
        model.read_f5c_model(bad)                                     # '#k' header is mandatory (src/model.c:96-99)


def test_worker_sharding_is_a_partition():
    for T, G in ((8, 2), (10, 4), (7, 8), (65536, 8), (5, 1)):
        seen = []
        for g in range(G):
            lo, hi = shard.worker_range(g, G, T)
            assert all(shard.owner_of(w, G, T) == g for w in range(lo, min(hi, lo + 5)))
            seen += list(range(lo, hi))
        assert seen == list(range(T))
    idx, wk = shard.shard_batch(10, 4, 1, 2)
    assert list(idx) == [6, 7, 8, 9] and list(wk) == [2, 2, 2, 3]


def test_read_ranges_partition_every_batch():
    """range sharding: the ranks' ranges are contiguous, disjoint, in rank order and cover the batch (also when there
    are fewer reads than ranks)"""
    from squigulator_amd import shard
    for n in (0, 1, 2, 7, 8, 9, 1000, 32768):
        for world in (1, 2, 3, 8):
            cuts = [shard.read_range(r, world, n) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
            assert all(lo <= hi for lo, hi in cuts)
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1
