#!/usr/bin/env python3
"""Sustained rate of the whole device-side flow from one host thread: batch i+1 is sampled and staged (device-side
gen_read, host descriptors) while batch i runs; results stay in HBM.  usage: python tools/pipeline.py [n_batches] [K]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from squigulator_amd import api, model, profiles
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 40
K = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
prof, fl = profiles.get_profile("dna-r9-prom")
mean, stdv = model.synthetic_model(6)
gen = api.SignalGenerator(prof, fl, 6, mean, stdv, 42, num_workers=K, mode=api.MODE_CERTIFIED)
gen.load_genome([bench.load_genome(bench.GENOME)], 10000, api.SAMPLE_DNA)
cur = gen.sample(K).run()
cur.wait(); cur.free()                                   # warm-up
cur = gen.sample(K).run()
samples = reads = 0
t0 = time.perf_counter()
for i in range(nb):
    nxt = gen.sample(K).run()                            # sampled, staged and queued while `cur` runs
    cur.wait()                                           # cur's results stay valid until two more batches have run
    samples += cur.n_samples; reads += cur.n_reads
    cur.free()
    cur = nxt
cur.wait(); samples += cur.n_samples; reads += cur.n_reads
dt = time.perf_counter() - t0
print(f"{nb + 1} batches of {K} reads, sampled on the device and generated, one host thread: {samples / dt:.3e} samples/s, "
      f"{reads / dt:.3e} reads/s, {1e3 * dt / (nb + 1):.2f} ms per batch")
