#!/usr/bin/env python3
"""End-to-end rates of the headline workload: the same batches with the output (i) left in HBM, (ii) fetched as
raw int16 into pinned host memory, (iii) written as BLOW5 (svb-zd on the device, framing + zlib on host threads) to a file
in /dev/shm.  One host thread drives everything; batch i+1 is queued before batch i is consumed.
usage: python tools/e2e_rates.py [reads_per_batch] [batches] [genome_mb]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402  (first: the HIP runtime torch brings is the one the library then uses)

torch.zeros(1, device="cuda")
import bench  # noqa: E402
from squigulator_amd import api, model, profiles  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 8
MB = float(sys.argv[3]) if len(sys.argv) > 3 else None
prof, fl = profiles.get_profile("dna-r10-prom")
mean, stdv = model.synthetic_model(9)
dev = torch.device("cuda", 0)
seq, lens = bench.synthetic_genome_device(MB, dev)


def run(kind):
    gen = api.SignalGenerator(prof, fl, 9, mean, stdv, 42, num_workers=1, mode=api.MODE_CERTIFIED)
    gen.load_genome_device(seq.data_ptr(), lens, 10000, api.SAMPLE_DNA)
    workers = np.zeros(K, np.int32)
    batches = [gen.sample(K, workers) for _ in range(NB + 1)]
    pinned = gen.pinned(2 * int(K * 10000 * 13 * 1.3), np.int16) if kind == "raw" else None
    w = api.Blow5Writer("/dev/shm/sqg_e2e.blow5", prof, fl, threads=0) if kind == "blow5" else None
    ids = [b"S1_%d!chr1!0!10000!+" % i for i in range(K)]
    batches[0].run().wait()                                # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    samples = 0
    batches[1].run()
    for i in range(1, NB + 1):
        if i + 1 <= NB:
            batches[i + 1].run()                           # the next batch computes while this one is consumed
        b = batches[i].wait()
        if kind == "raw":
            b.signal(out=pinned)
        elif kind == "blow5":
            w.write_batch(b, ids)
        samples += b.n_samples
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    size = w.close() if w else 0
    for b in batches:
        b.free()
    gen.close()
    return samples / dt, size, samples


for kind in ("discard", "raw", "blow5"):
    rate, size, ns = run(kind)
    extra = f", {size / ns:.3f} B/sample on disk" if size else ""
    print(f"{kind:8s} {rate:.3e} samples/s ({K} reads per batch, {NB} batches{extra})")
try:
    os.unlink("/dev/shm/sqg_e2e.blow5")
except OSError:
    pass
