#!/usr/bin/env python3
"""Generate golden vectors from the COMPILED REFERENCE path (oracle/_ref/ref_harness).

Runs only where /root/reference is mounted (this container).  For every case in
REFVEC_CASES it writes tests/golden/refvec/<id>.npz holding the reference's per-read outputs
(read coordinates, sequence, offset, median_before, int16 signal, per-event dwell) for a
SYNTHETIC pore model (squigulator_amd.model.synthetic_model), so the oracle -- and through it
the HIP path -- can be checked on machines without the reference.  Fixtures are data only.

usage: python tools/make_refvec.py            (rebuilds the harness first)
"""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from squigulator_amd import model, options  # noqa: E402
from refvec_cases import REFVEC_CASES  # noqa: E402

INPUTS = os.path.join(ROOT, "tests", "golden", "inputs")
OUT = os.path.join(ROOT, "tests", "golden", "refvec")
HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")

_model_files = {}


def model_file(k, tmp, meth=False):
    if (k, meth) not in _model_files:
        mean, stdv = model.synthetic_model(k, meth=meth)
        path = os.path.join(tmp, f"synthetic_{k}mer{'_meth' if meth else ''}.model")
        model.write_f5c_model(path, k, mean, stdv)
        _model_files[(k, meth)] = path
    return _model_files[(k, meth)]


def run_harness(cmdline, tmp, extra=None):
    o = options.parse_args(cmdline)
    k = o.kmer_size_default
    out = os.path.join(tmp, "out.bin")
    cfg = {
        "fasta": os.path.join(INPUTS, o.ref), "model": model_file(k, tmp, bool(o.meth_freq)), "out": out,
        "flags": o.flags, "amp_noise": repr(float(np.float32(o.amp_noise))), "seed": o.seed,
        "threads": o.threads, "batch": o.batch, "nreads": o.nreads, "rlen": o.rlen,
    }
    if o.trans_count:
        cfg["trans_count"] = os.path.join(INPUTS, o.trans_count)
    if o.meth_freq:
        cfg["meth_freq"] = os.path.join(INPUTS, o.meth_freq)
    for name, v in zip(("digitisation", "sample_rate", "bps", "range", "offset_mean", "offset_std",
                        "median_before_mean", "median_before_std", "dwell_mean", "dwell_std"),
                       o.profile.as_tuple()):
        cfg[name] = repr(float(v))
    if extra:
        cfg.update(extra)
    cfgp = os.path.join(tmp, "cfg.txt")
    with open(cfgp, "w") as f:
        for kk, v in cfg.items():
            f.write(f"{kk}={v}\n")
    subprocess.check_call([HARNESS, cfgp])
    return parse_dump(out), k


def parse_dump(path):
    with open(path, "rb") as f:
        buf = f.read()
    assert buf[:8] == b"SQGREF1\0"
    n = struct.unpack_from("<i", buf, 8)[0]
    p = 12
    reads = []
    for _ in range(n):
        tid, ref_idx, ref_len, pos_st, rlen, strand = struct.unpack_from("<6i", buf, p); p += 24
        offset, median = struct.unpack_from("<2d", buf, p); p += 16
        length, start_time, ssn = struct.unpack_from("<3q", buf, p); p += 24
        seq = buf[p:p + rlen]; p += rlen
        sig = np.frombuffer(buf, np.int16, length, p).copy(); p += 2 * length
        ss = np.frombuffer(buf, np.int32, ssn, p).copy(); p += 4 * ssn
        reads.append(dict(tid=tid, ref_idx=ref_idx, ref_len=ref_len, pos_st=pos_st, rlen=rlen,
                          strand=chr(strand), offset=offset, median=median, start_time=start_time,
                          seq=seq, sig=sig, ss=ss))
    assert p == len(buf)
    return reads


def pack(reads):
    meta = np.array([(r["tid"], r["ref_idx"], r["ref_len"], r["pos_st"], r["rlen"], ord(r["strand"]),
                      r["start_time"], len(r["sig"]), len(r["ss"])) for r in reads], np.int64)
    return dict(
        meta=meta,
        offset=np.array([r["offset"] for r in reads], np.float64),
        median=np.array([r["median"] for r in reads], np.float64),
        seq=np.frombuffer(b"".join(r["seq"] for r in reads), np.uint8),
        sig=np.concatenate([r["sig"] for r in reads]) if reads else np.zeros(0, np.int16),
        ss=np.concatenate([r["ss"] for r in reads]) if reads else np.zeros(0, np.int32),
    )


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    os.makedirs(OUT, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        for cid, cmd in REFVEC_CASES:
            reads, k = run_harness(cmd, tmp)
            np.savez_compressed(os.path.join(OUT, cid + ".npz"), cmd=np.array(cmd), k=np.array(k), **pack(reads))
            print(f"{cid}: {len(reads)} reads, {sum(len(r['sig']) for r in reads)} samples")


if __name__ == "__main__":
    main()
