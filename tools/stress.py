#!/usr/bin/env python3
"""Long-run self-consistency: certified vs exact mode, same seed, reads sampled on the device, N bench-sized batches;
every int16 compared.  usage: python tools/stress.py [n_batches] [profile]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from squigulator_amd import api, model, profiles
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 40
pname = sys.argv[2] if len(sys.argv) > 2 else "dna-r9-prom"
prof, fl = profiles.get_profile(pname)
k = profiles.default_kmer_size(fl)
mean, stdv = model.synthetic_model(k)
K = 8192
genome = bench.load_genome(bench.GENOME)
gens = [api.SignalGenerator(prof, fl, k, mean, stdv, 42, num_workers=K, mode=m) for m in (api.MODE_CERTIFIED, api.MODE_EXACT)]
for g in gens:
    g.load_genome([genome], 10000, api.SAMPLE_DNA)
bufs = None
total = 0; fb = 0; t0 = time.time()
for it in range(nb):
    bs = [g.sample(K).run() for g in gens]
    for b in bs:
        b.wait()
    fb += gens[0].timing()["fallback_samples"]
    if bufs is None or len(bufs[0]) < bs[0].n_samples:
        bufs = [np.empty(int(bs[0].n_samples * 1.2), np.int16) for _ in gens]
    sigs = [b.signal(buf) for b, buf in zip(bs, bufs)]
    assert bs[0].n_samples == bs[1].n_samples and np.array_equal(bs[0].sig_off, bs[1].sig_off), f"batch {it}: lengths differ"
    assert np.array_equal(sigs[0], sigs[1]), f"batch {it}: signals differ"
    total += bs[0].n_samples
    for b in bs:
        b.free()
print(f"{pname}: {nb} batches, {total:.3e} samples, certified == exact everywhere; {fb} samples ({fb / total:.2e}) took the FP64 fix-up; {time.time() - t0:.0f} s")
