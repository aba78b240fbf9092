#!/usr/bin/env python3
"""tests/golden/inputs/mfreq_dense.tsv: a methylation frequency for every fourth CpG of the nCoV genome (format of
--meth-freq: contig, 0-based position of the C, frequency), frequencies from a fixed generator incl. 0 and 1."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INP = os.path.join(ROOT, "tests", "golden", "inputs")
name, seq = None, []
for ln in open(os.path.join(INP, "nCoV-2019.reference.fasta")):
    if ln.startswith(">"):
        name = ln[1:].split()[0]
    else:
        seq.append(ln.strip())
seq = "".join(seq)
rng = np.random.default_rng(17)
cpg = [i for i in range(len(seq) - 1) if seq[i] == "C" and seq[i + 1] == "G"]
with open(os.path.join(INP, "mfreq_dense.tsv"), "w") as f:
    f.write("#chromosome\tposition\tfrequency\n")
    for j, i in enumerate(cpg):
        if j % 4 == 0:
            fr = [0.0, 1.0, 0.5][j // 4 % 3] if j % 28 == 0 else float(rng.random())
            f.write(f"{name}\t{i}\t{fr:.3f}\n")
print(len(cpg), "CpGs")
