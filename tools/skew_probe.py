import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np
import bench
from squigulator_amd import api, model, profiles
prefix = int(sys.argv[1])
prof, fl = profiles.get_profile("rna004-prom")
if prefix: fl |= profiles.SQ_PREFIX
mean, stdv = model.synthetic_model(9)
contigs = bench.load_contigs(bench.SEQUINS)
gen = api.SignalGenerator(prof, fl, 9, mean, stdv, 42, num_workers=1, mode=api.MODE_CERTIFIED)
gen.load_genome(contigs, 10000, api.SAMPLE_RNA)
K = 32768
bs = [gen.sample(K, np.zeros(K, np.int32)) for _ in range(3)]
for b in bs:
    b.run().wait()
    print(prefix, b.n_events, gen.timing())
