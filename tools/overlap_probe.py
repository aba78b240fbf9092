#!/usr/bin/env python3
"""How much throughput do concurrent streams buy?  Runs NCTX independent contexts (own stream each) on one GPU
from NCTX host threads and reports aggregate samples/s.  Exploration tool for the round-2 two-stream pipeline."""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from squigulator_amd import api, model, profiles

def main():
    nctx = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    steps = 8
    prof, fl = profiles.get_profile("dna-r9-prom")
    mean, stdv = model.synthetic_model(6)
    genome = bench.load_genome(bench.GENOME)
    gens, batches = [], []
    for c in range(nctx):
        g = api.SignalGenerator(prof, fl, 6, mean, stdv, seed=42 + c, num_workers=K, mode=api.MODE_CERTIFIED)
        rng = np.random.default_rng(c)
        bs = []
        for _ in range(steps + 1):
            blob, off = bench.pack(bench.sample_reads(genome, K, 10000, rng))
            bs.append(g.stage_packed(blob, off))
        bs[0].run().wait()
        gens.append(g); batches.append(bs[1:])
    tot = [0] * nctx
    def work(i):
        for b in batches[i]:
            b.run()
        for b in batches[i]:
            b.wait(); tot[i] += b.n_samples
    ths = [threading.Thread(target=work, args=(i,)) for i in range(nctx)]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    dt = time.perf_counter() - t0
    print(f"nctx={nctx} K={K}: {sum(tot) / dt:.3e} samples/s aggregate, {dt / steps * 1e3:.2f} ms per step-round")

main()
