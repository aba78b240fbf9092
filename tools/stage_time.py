#!/usr/bin/env python3
"""Host-side cost of getting a bench-sized batch ready: sqg_batch_stage (reads from the host) vs sqg_batch_sample
(reads drawn on the device-resident genome)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from squigulator_amd import api, model, profiles
prof, fl = profiles.get_profile("dna-r9-prom")
mean, stdv = model.synthetic_model(6)
KK = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
genome = bench.load_genome(bench.GENOME)
gen = api.SignalGenerator(prof, fl, 6, mean, stdv, 42, num_workers=KK, mode=api.MODE_CERTIFIED)
gen.load_genome([genome], 10000, api.SAMPLE_DNA)
rng = np.random.default_rng(1)
for it in range(3):
    blob, off = bench.pack(bench.sample_reads(genome, KK, 10000, rng))
    t0 = time.perf_counter(); b1 = gen.stage_packed(blob, off, None); t1 = time.perf_counter()
    b1.run().wait(); t2 = time.perf_counter()
    b2 = gen.sample(KK); t3 = time.perf_counter()
    b2.run().wait(); t4 = time.perf_counter()
    print(f"iter {it}: stage(host reads, {len(blob) / 1e6:.0f} MB) {1e3 * (t1 - t0):.1f} ms, run+wait {1e3 * (t2 - t1):.2f} ms | "
          f"sample(device) {1e3 * (t3 - t2):.1f} ms, run+wait {1e3 * (t4 - t3):.2f} ms")
    b1.free(); b2.free()
