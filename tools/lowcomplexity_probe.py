#!/usr/bin/env python3
"""Low-complexity genomes (a homopolymer, a dinucleotide repeat, a 50-base tandem repeat): every event of a batch falls on a handful of
k-mer streams -- the worst case for the in-order hand-out.  -t 1, 9-mer table, 8192 reads per batch: rate, and the ordered-LDS-atomic
kernels against the order-free ones.  usage: python tools/lowcomplexity_probe.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from squigulator_amd import api, model, profiles  # noqa: E402

prof, fl = profiles.get_profile("dna-r10-prom")
mean, stdv = model.synthetic_model(9)
rng = np.random.default_rng(3)
unit50 = bytes(rng.choice(list(b"ACGT"), 50).astype(np.uint8))
genomes = {"poly-A": b"A" * 4_000_000, "(AC)n": b"AC" * 2_000_000, "50-base tandem repeat": unit50 * 80_000}
K = 8192
w = np.zeros(K, np.int32)
for name, g in genomes.items():
    out = []
    for claims in (False, True):
        if claims:
            os.environ["SQG_PART_CLAIMS"] = "1"
        else:
            os.environ.pop("SQG_PART_CLAIMS", None)
        gen = api.SignalGenerator(prof, fl, 9, mean, stdv, 42, num_workers=1, mode=api.MODE_CERTIFIED)
        gen.load_genome([g], 10000, api.SAMPLE_DNA)
        tm = []
        sig = None
        for it in range(4):
            b = gen.sample(K, w).run()
            b.wait()
            tm.append(gen.timing())
            if it == 3:
                sig = b.signal().copy()
            n = b.n_samples
            b.free()
        gen.close()
        out.append((sig, tm[-1], n))
    eq = np.array_equal(out[0][0], out[1][0])
    t0, t1 = out[0][1], out[1][1]
    print(f"{name}: {out[0][2]:.3e} samples per batch; ordered: events {t0['events_ms']:.2f} ms, total {t0['total_ms']:.2f} ms "
          f"({out[0][2] / t0['total_ms'] / 1e9 * 1e3:.0f} Gsamples/s); order-free: events {t1['events_ms']:.2f} ms; equal: {eq}")
