#!/usr/bin/env python3
"""Golden vectors for the svb-zd signal compression ("next" row) from the REFERENCE's own slow5lib
(oracle/_ref/ref_harness, svb_in/svb_out mode = slow5_ptr_compress_solo(SLOW5_COMPRESS_SVB_ZD, ...)).

Runs only where /root/reference is mounted.  Writes tests/golden/svb/svb_cases.npz: a set of int16 arrays
(simulated signals from the committed refvec fixtures plus adversarial ones: empty, length 1..9, constant,
full-range jumps that need 3-byte codes, random) and the bytes the library produces for each.  Data only.
usage: python tools/make_svbvec.py"""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
OUT = os.path.join(ROOT, "tests", "golden", "svb")


def cases():
    rng = np.random.default_rng(12345)
    out = [np.zeros(0, np.int16)]
    for n in range(1, 10):
        out.append(rng.integers(-300, 300, n).astype(np.int16))
    out.append(np.full(1000, 517, np.int16))
    out.append(np.array([-32768, 32767] * 37 + [0, -1, 1, 127, 128, -128, -129, 255, 256, 32767, -32768], np.int16))   # 3-byte codes
    out.append(rng.integers(-32768, 32768, 4099).astype(np.int16))
    out.append((500 + 40 * rng.standard_normal(100003)).astype(np.int16))
    for name in ("r9_t1", "rna004_prefix"):
        p = os.path.join(ROOT, "tests", "golden", "refvec", name + ".npz")
        if os.path.exists(p):
            d = np.load(p)
            out.append(d["sig"][:200000].astype(np.int16))
    return out


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    os.makedirs(OUT, exist_ok=True)
    arrs = cases()
    with tempfile.TemporaryDirectory() as tmp:
        fin, fout, cfg = (os.path.join(tmp, x) for x in ("in.bin", "out.bin", "cfg.txt"))
        with open(fin, "wb") as f:
            f.write(struct.pack("<i", len(arrs)))
            for a in arrs:
                f.write(struct.pack("<q", len(a)))
                f.write(a.tobytes())
        with open(cfg, "w") as f:
            f.write(f"svb_in={fin}\nsvb_out={fout}\n")
        subprocess.check_call([HARNESS, cfg])
        buf = open(fout, "rb").read()
    p = 0
    enc = []
    for _ in arrs:
        (nb,) = struct.unpack_from("<q", buf, p); p += 8
        enc.append(np.frombuffer(buf, np.uint8, nb, p).copy()); p += nb
    assert p == len(buf)
    np.savez_compressed(os.path.join(OUT, "svb_cases.npz"),
                        lens=np.array([len(a) for a in arrs], np.int64), sig=np.concatenate(arrs),
                        enc_lens=np.array([len(e) for e in enc], np.int64), enc=np.concatenate(enc))
    print(f"{len(arrs)} arrays, {sum(len(a) for a in arrs)} samples -> {sum(len(e) for e in enc)} bytes")


if __name__ == "__main__":
    main()
