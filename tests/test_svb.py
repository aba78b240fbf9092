"""svb-zd signal compression ("next" row of SURVEY.md section 8f): slow5lib/src/slow5_press.c:1055-1087.

CPU: the oracle restatement against bytes produced by the reference's own slow5lib (tests/golden/svb, made by
tools/make_svbvec.py through oracle/_ref/ref_harness), and the round trip.  GPU: the device coder against the
oracle on simulated batches, plus the decode round trip at full size."""
import os

import numpy as np
import pytest

import orc

GOLD = os.path.join(os.path.dirname(__file__), "golden", "svb", "svb_cases.npz")


def _cases():
    d = np.load(GOLD)
    so = np.concatenate(([0], np.cumsum(d["lens"])))
    eo = np.concatenate(([0], np.cumsum(d["enc_lens"])))
    for i in range(len(d["lens"])):
        yield d["sig"][so[i]:so[i + 1]], d["enc"][eo[i]:eo[i + 1]]


def test_oracle_svb_zd_equals_slow5lib_bytes():
    n = 0
    for sig, want in _cases():
        got = orc.svb_zd(sig)
        np.testing.assert_array_equal(got, want, err_msg=f"array of {len(sig)} samples")
        dec, used = orc.svb_zd_decode(got)
        np.testing.assert_array_equal(dec, sig)
        assert used == len(got)
        n += 1
    assert n >= 10


def test_live_against_slow5lib_when_reference_is_mounted(tmp_path):
    """Randomised arrays through the compiled reference (only where /root/reference exists)."""
    import struct
    import subprocess
    harness = os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "ref_harness")
    if not (os.path.isdir("/root/reference") and os.path.exists(harness)):
        pytest.skip("compiled reference not available")
    rng = np.random.default_rng(7)
    arrs = [rng.integers(-32768, 32768, int(n)).astype(np.int16) for n in rng.integers(0, 5000, 20)]
    arrs += [(rng.integers(200, 900) + 50 * rng.standard_normal(int(n))).astype(np.int16) for n in rng.integers(1, 50000, 10)]
    fin, fout, cfg = (str(tmp_path / x) for x in ("in.bin", "out.bin", "cfg.txt"))
    with open(fin, "wb") as f:
        f.write(struct.pack("<i", len(arrs)))
        for a in arrs:
            f.write(struct.pack("<q", len(a)) + a.tobytes())
    open(cfg, "w").write(f"svb_in={fin}\nsvb_out={fout}\n")
    subprocess.check_call([harness, cfg])
    buf = open(fout, "rb").read()
    p = 0
    for a in arrs:
        (nb,) = struct.unpack_from("<q", buf, p); p += 8
        np.testing.assert_array_equal(orc.svb_zd(a), np.frombuffer(buf, np.uint8, nb, p)); p += nb


@pytest.mark.gpu
@pytest.mark.parametrize("profile,extra", [("dna-r9-prom", 0), ("rna004-prom", 0x20), ("dna-r10-prom", 0)],
                         ids=["r9", "rna004_prefix", "r10"])
def test_device_svb_zd_equals_oracle(profile, extra):
    from squigulator_amd import api, model, profiles
    prof, fl = profiles.get_profile(profile)
    fl |= extra
    k = 9 if ("r10" in profile or "rna004" in profile) else 6
    mean, stdv = model.synthetic_model(k)
    rng = np.random.default_rng(5)
    seqs = [bytes(rng.choice(list(b"ACGT"), int(n)).astype(np.uint8)) for n in (3, 7, 250, 1000, 2500, 1, 800, 4097)]
    gen = api.SignalGenerator(prof, fl, k, mean, stdv, 42, num_workers=len(seqs), mode=api.MODE_CERTIFIED)
    b = gen.submit(seqs)
    sig = b.signal()
    enc, off = b.compress()
    assert off[0] == 0 and off[-1] == len(enc)
    for i in range(len(seqs)):
        s = sig[b.sig_off[i]:b.sig_off[i + 1]]
        np.testing.assert_array_equal(enc[off[i]:off[i + 1]], orc.svb_zd(s), err_msg=f"read {i} ({len(s)} samples)")
    b.free()
    e = gen.submit([])
    enc, off = e.compress()
    assert len(enc) == 0 and list(off) == [0]
    e.free()
    gen.close()


@pytest.mark.gpu
def test_device_svb_zd_round_trip_full_size():
    """A bench-sized batch: every read's encoding decodes back to its signal (size-independent property), and
    a sample of reads equals the oracle's bytes."""
    import bench
    from squigulator_amd import api, model, profiles
    prof, fl = profiles.get_profile("dna-r9-prom")
    mean, stdv = model.synthetic_model(6)
    genome = bench.load_genome(bench.GENOME)
    reads = bench.sample_reads(genome, 2048, 10000, np.random.default_rng(3))
    gen = api.SignalGenerator(prof, fl, 6, mean, stdv, 42, num_workers=len(reads), mode=api.MODE_CERTIFIED)
    b = gen.submit(reads)
    sig = b.signal()
    enc, off = b.compress()
    assert len(enc) < 0.8 * 2 * len(sig)                     # it compresses (about 1.3 B per sample)
    for i in range(0, len(reads), 37):
        s = sig[b.sig_off[i]:b.sig_off[i + 1]]
        dec, used = orc.svb_zd_decode(enc[off[i]:off[i + 1]])
        np.testing.assert_array_equal(dec, s)
        assert used == off[i + 1] - off[i]
    for i in (0, 1, len(reads) - 1):
        np.testing.assert_array_equal(enc[off[i]:off[i + 1]], orc.svb_zd(sig[b.sig_off[i]:b.sig_off[i + 1]]))
    b.free(); gen.close()
