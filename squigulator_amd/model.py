"""Pore-model tables: f5c-format text I/O and the synthetic stand-in tables.

The reference's built-in ONT tables (src/model.h, src/methmodel.c) are not part
of the mounted tree, so `-x` presets here use clearly-labelled SYNTHETIC tables
from a fixed formula: level_mean in [60,140), level_stdv in
[1,4), both multiples of 1/64 so the text form ("%f") and the float32 value
are exactly the same number.  Throughput does not depend on table values.
File format: src/model.c:40-142 ('#k\\t<k>' header line mandatory).
"""
from __future__ import annotations

import numpy as np


def synthetic_model(k: int, salt: int = 0, meth: bool = False):
    """(level_mean float32[n], level_stdv float32[n]) from a fixed integer hash of the rank; n = 4^k, or 5^k for the
    5-letter (A C G M T) methylation tables (src/sim.c:325)."""
    n = 5 ** k if meth else 1 << (2 * k)
    r = np.arange(n, dtype=np.uint64)
    h = (r + np.uint64(salt) * np.uint64(0x9E3779B9)) & np.uint64(0xFFFFFFFF)
    h = (h * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(2246822519)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(3266489917)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    mean = 60.0 + (h % np.uint64(5120)).astype(np.float64) / 64.0
    stdv = 1.0 + ((h >> np.uint64(13)) % np.uint64(192)).astype(np.float64) / 64.0
    return mean.astype(np.float32), stdv.astype(np.float32)


_BASES = "ACGT"


def kmer_string(rank: int, k: int) -> str:
    return "".join(_BASES[(rank >> (2 * (k - 1 - i))) & 3] for i in range(k))


def meth_kmer_string(rank: int, k: int) -> str:
    """the k-mer of a 5-letter rank (src/seq.h:45-74: A C G M T = 0..4, first base most significant)"""
    return "".join("ACGMT"[(rank // 5 ** (k - 1 - i)) % 5] for i in range(k))


def write_f5c_model(path, k: int, mean, stdv) -> None:
    """f5c-format text table; 5^k rows are written with the 5-letter k-mers (rows are taken in file order, src/model.c:100)"""
    meth = len(mean) == 5 ** k
    with open(path, "w") as f:
        f.write("#model_name\tsynthetic\n")
        f.write(f"#k\t{k}\n")
        f.write("kmer\tlevel_mean\tlevel_stdv\tsd_mean\tsd_stdv\n")
        for r in range(len(mean)):
            f.write(f"{meth_kmer_string(r, k) if meth else kmer_string(r, k)}\t{float(mean[r]):f}\t{float(stdv[r]):f}\t0.0\t0.0\n")


def read_f5c_model(path):
    """-> (k, level_mean float32[], level_stdv float32[]); values via strtof-equivalent parsing."""
    k = 0
    means, stdvs = [], []
    with open(path) as f:
        for line in f:
            if line.startswith("#"):
                parts = line.split()
                if len(parts) == 2 and parts[0] == "#k":
                    k = int(parts[1])
                    if not 0 < k <= 9:
                        raise ValueError(f"invalid k-mer size {k} in {path}")
                continue
            if line.startswith("kmer") or line in ("\n", "\r\n"):
                continue
            if not k:
                raise ValueError(f"Invalid model file {path}: '#k' header is missing")
            p = line.split()
            if len(p) < 3 or len(p[0]) != k:
                raise ValueError(f"{path}: k-mer size inconsistent with header ({k})")
            means.append(np.float32(p[1]))
            stdvs.append(np.float32(p[2]))
    if len(means) != 1 << (2 * k):
        raise ValueError(f"{path}: expected {1 << (2 * k)} k-mers, found {len(means)}")
    return k, np.asarray(means, np.float32), np.asarray(stdvs, np.float32)
