#!/usr/bin/env python3
"""Few workers, bench-sized batches: the hand-out by ordered LDS atomics (k_part_events, k_part_hand_ord) against the order-free
kernels (SQG_PART_CLAIMS=1: k_events<..,PART> / k_link_prefix, k_part_hand) -- two independent implementations, same seed, reads
sampled on the device, every int16 compared.  usage: python tools/stress_few.py [n_batches] [profile] [workers] [reads per batch]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from squigulator_amd import api, model, profiles  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 10
pname = sys.argv[2] if len(sys.argv) > 2 else "dna-r10-prom"
W = int(sys.argv[3]) if len(sys.argv) > 3 else 1
K = int(sys.argv[4]) if len(sys.argv) > 4 else 16384
prof, fl = profiles.get_profile(pname)
rna = bool(fl & profiles.SQ_RNA)
if rna:
    fl |= profiles.SQ_PREFIX
k = profiles.default_kmer_size(fl)
mean, stdv = model.synthetic_model(k)
contigs = bench.load_contigs(bench.SEQUINS) if rna else [bench.load_genome(bench.GENOME)]
gens = []
for claims in (False, True):
    if claims:
        os.environ["SQG_PART_CLAIMS"] = "1"
    g = api.SignalGenerator(prof, fl, k, mean, stdv, 42, num_workers=W, mode=api.MODE_CERTIFIED)
    g.load_genome(contigs, 10000, api.SAMPLE_RNA if rna else api.SAMPLE_DNA)
    assert g.probe_lds_order(64, 2)[1] == (not claims)
    gens.append(g)
workers = (np.arange(K) % W).astype(np.int32)
bufs = None
total = 0
t0 = time.time()
for it in range(nb):
    bs = [g.sample(K, workers).run() for g in gens]
    for b in bs:
        b.wait()
    if bufs is None or len(bufs[0]) < bs[0].n_samples:
        bufs = [np.empty(int(bs[0].n_samples * 1.2), np.int16) for _ in gens]
    sigs = [b.signal(buf) for b, buf in zip(bs, bufs)]
    assert bs[0].n_samples == bs[1].n_samples and np.array_equal(bs[0].sig_off, bs[1].sig_off), f"batch {it}: lengths differ"
    assert np.array_equal(sigs[0], sigs[1]), f"batch {it}: signals differ"
    total += bs[0].n_samples
    for b in bs:
        b.free()
print(f"{pname} -t {W} -K {K}: {nb} batches, {total:.3e} samples, ordered == order-free everywhere; {time.time() - t0:.0f} s")
