#!/usr/bin/env python3
"""PCIe-inclusive rates of one bench-sized batch: raw int16 fetch vs svb-zd on the device + fetch."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from squigulator_amd import api, model, profiles
prof, fl = profiles.get_profile("dna-r9-prom")
mean, stdv = model.synthetic_model(6)
gen = api.SignalGenerator(prof, fl, 6, mean, stdv, 42, num_workers=8192, mode=api.MODE_CERTIFIED)
gen.load_genome([bench.load_genome(bench.GENOME)], 10000, api.SAMPLE_DNA)
for it in range(3):
    b = gen.sample(8192)
    t0 = time.perf_counter(); b.run().wait(); t1 = time.perf_counter()
    sig = b.signal(); t2 = time.perf_counter()
    enc, off = b.compress(); t3 = time.perf_counter()
    n = b.n_samples
    print(f"batch {it}: {n} samples; generate {1e3 * (t1 - t0):.2f} ms; raw fetch {1e3 * (t2 - t1):.1f} ms "
          f"({2 * n / (t2 - t1) / 1e9:.1f} GB/s) -> {n / (t2 - t0):.3e} samples/s incl. PCIe; "
          f"svb-zd compress+fetch {1e3 * (t3 - t2):.1f} ms ({len(enc) / n:.3f} B/sample) -> {n / ((t1 - t0) + (t3 - t2)):.3e} samples/s incl. PCIe")
    b.free()
