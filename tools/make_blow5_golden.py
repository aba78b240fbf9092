#!/usr/bin/env python3
"""Golden BLOW5 files from the COMPILED REFERENCE path: oracle/_ref/ref_harness runs the reference's gen_read/gen_sig and
writes the records through the reference's own slow5lib (slow5_open "w" on a .blow5 path: zlib + svb-zd), with the header
setters of src/gensig.c.  Runs only where /root/reference is mounted.  For every case of BLOW5_CASES it writes
tests/golden/blow5/<id>.blow5 (data: what `squigulator -o x.blow5` writes for that command line with the synthetic pore table)
and <id>.npz with the per-read inputs a writer needs (read ids, offsets, medians, signals).

usage: python tools/make_blow5_golden.py"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import make_refvec as mr  # noqa: E402
from blow5_cases import BLOW5_CASES  # noqa: E402
from squigulator_amd import options, slow5_text  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "blow5")


def ref_names(fasta):
    return [ln[1:].split()[0] for ln in open(fasta) if ln.startswith(">")]


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    os.makedirs(OUT, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        for cid, cmd in BLOW5_CASES:
            b5 = os.path.join(OUT, cid + ".blow5")
            reads, k = mr.run_harness(cmd, tmp, extra={"slow5": b5})
            o = options.parse_args(cmd)
            names = ref_names(os.path.join(mr.INPUTS, o.ref))
            ids = [slow5_text.read_id(o.flags, i + 1, names[r["ref_idx"]], r["pos_st"], r["pos_st"] + r["rlen"], r["strand"]).encode()
                   for i, r in enumerate(reads)]
            np.savez_compressed(os.path.join(OUT, cid + ".npz"), cmd=np.array(cmd),
                                ids=np.frombuffer(b"\n".join(ids), np.uint8),
                                offset=np.array([r["offset"] for r in reads]), median=np.array([r["median"] for r in reads]),
                                lens=np.array([len(r["sig"]) for r in reads], np.int64),
                                sig=np.concatenate([r["sig"] for r in reads]))
            print(f"{cid}: {len(reads)} reads, {os.path.getsize(b5)} bytes")


if __name__ == "__main__":
    main()
