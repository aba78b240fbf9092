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


def test_meth_freq_table_is_validated_like_the_reference(tmp_path):
    """load_meth_freq, src/ref.c:314-345: unknown contig, negative / out-of-range position, not a C, frequency outside [0, 1],
    missing columns -- each is an error naming the line (the reference exits), none reaches the uint8 table"""
    import pytest
    from squigulator_amd import api
    contigs, names = [b"ACGTCCGA", b"ggcatc"], ["chr1", "chr2"]

    def run(text):
        p = tmp_path / "m.tsv"
        p.write_text(text)
        return api.load_meth_freq(contigs, names, str(p))

    blob, has = run("#chr\tpos\tfreq\nchr1\t1\t0.5\nchr2\t2\t1\nchr1\t4\t0\n")
    assert list(has) == [1, 1] and blob[1] == 128 and blob[8 + 2] == 255 and blob[4] == 0 and blob.sum() == 128 + 255
    for text, what in (("chrX\t1\t0.5\n", "no such chromosome"), ("chr1\t-1\t0.5\n", "cannot be negative"),
                       ("chr1\t8\t0.5\n", "must be less than the length 8"), ("chr1\t0\t0.5\n", "was a A"),
                       ("chr1\t1\t1.5\n", "between 0 to 1"), ("chr1\t1\t-0.1\n", "between 0 to 1"), ("chr1\t1\n", "malformed line"),
                       ("chr1\n", "malformed line")):
        with pytest.raises(ValueError, match=what):
            run(text)


def test_blow5_writer_reports_io_errors(tmp_path):
    """sqg_blow5_open on a path that cannot be created: SQG_EIO (not a generic SQG_EINVAL), the reason on stderr"""
    import pytest
    from squigulator_amd import api, profiles
    prof, fl = profiles.get_profile("dna-r9-prom")
    with pytest.raises(api.SqgError) as ei:
        api.Blow5Writer(str(tmp_path / "no" / "such" / "dir" / "x.blow5"), prof, fl)
    assert ei.value.code == -7
    w = api.Blow5Writer(str(tmp_path / "ok.blow5"), prof, fl)
    assert w.close() > 0
