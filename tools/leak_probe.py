#!/usr/bin/env python3
"""create / use / destroy many contexts (streamed batches, so that precount, the draw-ahead thread and the sampler's pinned buffer are all
in play): device memory and host threads must come back.  usage: python tools/leak_probe.py [contexts]"""
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

torch.zeros(1, device="cuda")
import bench  # noqa: E402
from squigulator_amd import api, model, profiles  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
prof, fl = profiles.get_profile("dna-r10-prom")
mean, stdv = model.synthetic_model(9)
contigs = bench.synthetic_genome_host(8.0)
free0 = None
for it in range(N):
    gen = api.SignalGenerator(prof, fl, 9, mean, stdv, 42 + it, num_workers=1 + it % 3, mode=api.MODE_CERTIFIED)
    gen.load_genome(contigs, 3000, api.SAMPLE_DNA)
    cur = gen.sample(300).run()
    nxt = gen.sample(300)
    for _ in range(4):
        nn = gen.sample(300)
        nxt.run(); cur.wait(); cur.free(); cur, nxt = nxt, nn
    cur.wait(); cur.free(); nxt.free()
    gen.close()
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    if it == 2:
        free0, thr0 = free, threading.active_count()
    if it % 10 == 9:
        print(f"context {it + 1}: free device memory {free / 2**20:.0f} MiB, process threads {len(os.listdir('/proc/self/task'))}")
print("device memory drift since context 3: %.1f MiB" % ((free0 - free) / 2**20))
assert free0 - free < 64 * 2**20, "device memory does not come back"
