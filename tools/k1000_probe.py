#!/usr/bin/env python3
"""Where a 1000-read batch's host time goes in the streaming pattern (stage i+2, run i+1, wait i, free i): per-call wall times.
usage: python tools/k1000_probe.py [reads_per_batch] [batches] [genome_mb] [workers]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

torch.zeros(1, device="cuda")
import bench  # noqa: E402
from squigulator_amd import api, model, profiles  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 400
MB = float(sys.argv[3]) if len(sys.argv) > 3 else 256.0
T = int(sys.argv[4]) if len(sys.argv) > 4 else 1
prof, fl = profiles.get_profile("dna-r10-prom")
mean, stdv = model.synthetic_model(9)
dev = torch.device("cuda", 0)
seq, lens = bench.synthetic_genome_device(MB, dev)
gen = api.SignalGenerator(prof, fl, 9, mean, stdv, 42, num_workers=T, mode=api.MODE_CERTIFIED)
gen.load_genome_device(seq.data_ptr(), lens, 10000, api.SAMPLE_DNA)
gen.set_phase_timing(0)
workers = (np.arange(K, dtype=np.int32) // max(K // T, 1)).clip(0, T - 1).astype(np.int32)
t = {"sample": 0.0, "run": 0.0, "wait": 0.0, "free": 0.0}
ns = 0
t0 = time.perf_counter()
cur = gen.sample(K, workers).run()
nxt = gen.sample(K, workers)
for it in range(NB + 20):
    if it == 20:
        torch.cuda.synchronize(); t = dict.fromkeys(t, 0.0); t0 = time.perf_counter(); ns = 0
    a = time.perf_counter(); nn = gen.sample(K, workers)
    b = time.perf_counter(); nxt.run()
    c = time.perf_counter(); cur.wait()
    d = time.perf_counter(); ns += cur.n_samples if it >= 20 else 0; cur.free()
    e = time.perf_counter()
    t["sample"] += b - a; t["run"] += c - b; t["wait"] += d - c; t["free"] += e - d
    cur, nxt = nxt, nn
tot = time.perf_counter() - t0
print(f"{K} reads per batch, -t {T}: {tot / NB * 1e3:.3f} ms per batch, {ns / tot:.3e} samples/s; per call (ms): " + ", ".join(f"{k} {v / NB * 1e3:.3f}" for k, v in t.items()))
