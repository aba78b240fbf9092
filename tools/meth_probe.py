#!/usr/bin/env python3
"""Throughput of the 5-letter (methylation) tables next to the 4-letter ones on the same reads: dna-r9-prom, 8192 reads of 10 kb,
`-t 1` and `T = K`.   python tools/meth_probe.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from squigulator_amd import api, model, profiles  # noqa: E402

K, L = 8192, 10000
rng = np.random.default_rng(1)
flat = rng.choice(np.frombuffer(b"ACGT", np.uint8), K * L)
meth = flat.copy()
cpg = np.flatnonzero((flat[:-1] == ord("C")) & (flat[1:] == ord("G")))
meth[cpg[rng.random(len(cpg)) < 0.7]] = ord("M")
for name, arr, fl_extra in (("4-letter", flat, 0), ("5-letter, 70 % of the CpGs methylated", meth, profiles.SQ_METH)):
    reads = [arr[i * L:(i + 1) * L].tobytes() for i in range(K)]
    prof, fl = profiles.get_profile("dna-r9-prom")
    fl |= fl_extra
    k = 6
    mean, stdv = model.synthetic_model(k, meth=bool(fl_extra))
    for T in (1, K):
        gen = api.SignalGenerator(prof, fl, k, mean, stdv, 42, num_workers=T, mode=api.MODE_CERTIFIED)
        staged = [gen.stage(reads) for _ in range(8)]
        for b in staged[:2]:
            b.run().wait()
        t0 = time.perf_counter()
        for b in staged[2:]:
            b.run()
        n = 0
        ev = []
        for b in staged[2:]:
            b.wait(); n += b.n_samples; ev.append(gen.timing()["events_ms"])
        dt = time.perf_counter() - t0
        print(f"{name:42s} -t {T:<5d}: {n / dt:.3e} samples/s, {dt / 6 * 1e3:.2f} ms per batch, event side {np.mean(ev):.2f} ms")
        for b in staged:
            b.free()
        gen.close()
