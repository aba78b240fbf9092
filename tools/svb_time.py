#!/usr/bin/env python3
"""Wall time of the device svb-zd coder on a bench-sized batch (blocking call, no D2H)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from squigulator_amd import api, model, profiles
prof, fl = profiles.get_profile("dna-r9-prom")
mean, stdv = model.synthetic_model(6)
reads = bench.sample_reads(bench.load_genome(bench.GENOME), 8192, 10000, np.random.default_rng(3))
gen = api.SignalGenerator(prof, fl, 6, mean, stdv, 42, num_workers=len(reads), mode=api.MODE_CERTIFIED)
for it in range(3):
    b = gen.submit(reads)
    t0 = time.perf_counter(); _, off = b.compress(fetch=False); dt = time.perf_counter() - t0
    print(f"batch {it}: {b.n_samples} samples -> {off[-1]} bytes ({off[-1] / b.n_samples:.3f} B/sample), compress {dt * 1e3:.3f} ms "
          f"= {b.n_samples / dt:.3e} samples/s")
    b.free()
