#!/usr/bin/env python3
"""-t 8 -K 1000 streaming from a SECOND context of the process (bench.py's small_batch leg) against the same from the only context:
usage: python tools/two_ctx_probe.py [genome_mb] [keep_first 0/1] [first_ran_big 0/1]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

torch.zeros(1, device="cuda")
import bench  # noqa: E402
from squigulator_amd import api, model, profiles  # noqa: E402

MB = float(sys.argv[1]) if len(sys.argv) > 1 else 3088.0
KEEP = int(sys.argv[2]) if len(sys.argv) > 2 else 1
BIG = int(sys.argv[3]) if len(sys.argv) > 3 else 0
prof, fl = profiles.get_profile("dna-r10-prom")
mean, stdv = model.synthetic_model(9)
dev = torch.device("cuda", 0)
seq, lens = bench.synthetic_genome_device(MB, dev)
g1 = api.SignalGenerator(prof, fl, 9, mean, stdv, 42, num_workers=1, mode=api.MODE_CERTIFIED)
g1.load_genome_device(seq.data_ptr(), lens, 10000, api.SAMPLE_DNA)
if BIG:
    b = g1.sample(16384, np.zeros(16384, np.int32)).run(); b.wait(); b.free()
if not KEEP:
    g1.close()
g2 = api.SignalGenerator(prof, fl, 9, mean, stdv, 42, num_workers=8, mode=api.MODE_CERTIFIED)
g2.load_genome_device(seq.data_ptr(), lens, 10000, api.SAMPLE_DNA)
g2.set_phase_timing(0)
wk = (np.arange(1000, dtype=np.int32) // 125).astype(np.int32)
r = bench.pipeline_leg(lambda: g2.sample(1000, wk), lambda b: b.run(), 64)
r = bench.pipeline_leg(lambda: g2.sample(1000, wk), lambda b: b.run(), 1500)
print(f"genome {MB:.0f} MB, first context kept {KEEP}, ran a 16384-read batch {BIG}: {r[2] / 1501 * 1e3:.3f} ms per batch, host staging {r[3] * 1e3:.3f} ms, {r[0] / r[2]:.3e} samples/s")
