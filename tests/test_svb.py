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
