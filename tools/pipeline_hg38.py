#!/usr/bin/env python3
"""Sustained rate of the headline workload with NOTHING staged ahead: one host thread samples and stages batch i+1 (device-side
gen_read on the 3.09 Gb genome, descriptors, links and groups) while batch i runs; results stay in HBM.
usage: python tools/pipeline_hg38.py [n_batches] [K] [genome_mb]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

torch.zeros(1, device="cuda")
import bench  # noqa: E402
from squigulator_amd import api, model, profiles  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 20
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
MB = float(sys.argv[3]) if len(sys.argv) > 3 else None
prof, fl = profiles.get_profile("dna-r10-prom")
mean, stdv = model.synthetic_model(9)
seq, lens = bench.synthetic_genome_device(MB, torch.device("cuda", 0))
gen = api.SignalGenerator(prof, fl, 9, mean, stdv, 42, num_workers=1, mode=api.MODE_CERTIFIED)
gen.load_genome_device(seq.data_ptr(), lens, 10000, api.SAMPLE_DNA)
workers = np.zeros(K, np.int32)
cur = gen.sample(K, workers).run()
cur.wait(); cur.free()
t_stage = []
cur = gen.sample(K, workers).run()
samples = reads = 0
t0 = time.perf_counter()
for i in range(nb):
    a = time.perf_counter()
    nxt = gen.sample(K, workers)
    t_stage.append(time.perf_counter() - a)
    nxt.run()
    cur.wait()
    samples += cur.n_samples; reads += cur.n_reads
    cur.free()
    cur = nxt
cur.wait(); samples += cur.n_samples; reads += cur.n_reads
dt = time.perf_counter() - t0
print(f"{nb + 1} batches of {K} reads (-t 1, dna-r10-prom, {sum(lens) / 1e9:.2f} Gb genome), sampled, staged and generated from one host "
      f"thread: {samples / dt:.3e} samples/s, {reads / dt:.3e} reads/s, {1e3 * dt / (nb + 1):.2f} ms per batch; host time in "
      f"sqg_batch_sample {1e3 * np.mean(t_stage):.2f} ms per batch")
