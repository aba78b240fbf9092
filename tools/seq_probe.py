import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
torch.zeros(1, device="cuda")
import bench
from squigulator_amd import api, model, profiles
prof, fl = profiles.get_profile("dna-r10-prom")
mean, stdv = model.synthetic_model(9)
seq, lens = bench.synthetic_genome_device(None, torch.device("cuda", 0))
gen = api.SignalGenerator(prof, fl, 9, mean, stdv, 42, num_workers=1, mode=api.MODE_CERTIFIED)
gen.load_genome_device(seq.data_ptr(), lens, 10000, api.SAMPLE_DNA)
K = 16384
w = np.zeros(K, np.int32)
bs = [gen.sample(K, w) for _ in range(8)]
ev = []; ln = []; tot = []
for b in bs:
    b.run().wait(); torch.cuda.synchronize()
    t = gen.timing(); ev.append(t["events_ms"]); ln.append(t["lean_ms"]); tot.append(t["total_ms"])
print("one batch at a time: events %.3f lean %.3f total %.3f" % (np.mean(ev[2:]), np.mean(ln[2:]), np.mean(tot[2:])))
