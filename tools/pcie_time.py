#!/usr/bin/env python3
"""PCIe-inclusive rates of one bench-sized batch: raw int16 fetch vs svb-zd on the device + fetch, into pageable
and into pinned (sqg_host_alloc) host memory."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from squigulator_amd import api, model, profiles
prof, fl = profiles.get_profile("dna-r9-prom")
mean, stdv = model.synthetic_model(6)
gen = api.SignalGenerator(prof, fl, 6, mean, stdv, 42, num_workers=8192, mode=api.MODE_CERTIFIED)
gen.load_genome([bench.load_genome(bench.GENOME)], 10000, api.SAMPLE_DNA)
pin16 = gen.pinned(2 * 700_000_000, np.int16)
pin8 = gen.pinned(1_000_000_000, np.uint8)
page16 = np.empty(700_000_000, np.int16); page16[:] = 0            # touched once: no first-touch faults in the timing
for it in range(3):
    b = gen.sample(8192)
    t0 = time.perf_counter(); b.run().wait(); t1 = time.perf_counter()
    n = b.n_samples
    ta = time.perf_counter(); b.signal(page16); tb = time.perf_counter()
    b.signal(pin16); tc = time.perf_counter()
    enc, off = b.compress(out=pin8); td = time.perf_counter()
    gen_s = t1 - t0
    print(f"batch {it}: {n} samples, generate {1e3 * gen_s:.2f} ms | raw->pageable {1e3 * (tb - ta):.1f} ms ({2 * n / (tb - ta) / 1e9:.1f} GB/s) "
          f"=> {n / (gen_s + tb - ta):.2e} samples/s | raw->pinned {1e3 * (tc - tb):.1f} ms ({2 * n / (tc - tb) / 1e9:.1f} GB/s) => "
          f"{n / (gen_s + tc - tb):.2e} | svb-zd ({len(enc) / n:.3f} B/sample) compress+fetch->pinned {1e3 * (td - tc):.1f} ms => {n / (gen_s + td - tc):.2e}")
    b.free()
