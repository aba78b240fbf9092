#!/usr/bin/env python3
"""The reference's few-worker regimes (`-t 1`, `-t 8`): a worker's chain of reads cut into links (default) against the
serial walk (SQG_SPLIT_CHAINS=0).  usage: python tools/few_workers.py [T] [K] [workload]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from squigulator_amd import api, model, profiles
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
wl = sys.argv[3] if len(sys.argv) > 3 else "ncov-r9"
pname, wflags, wmode = bench.WORKLOADS[wl][:3]
prof, fl = profiles.get_profile(pname)
fl |= wflags
k = profiles.default_kmer_size(fl)
mean, stdv = model.synthetic_model(k)
contigs = bench.synthetic_genome_host(64) if wl == "synth-r10" else [bench.load_genome(bench.GENOME)]
for setting in ("0", None):
    if setting is None:
        os.environ.pop("SQG_SPLIT_CHAINS", None)
    else:
        os.environ["SQG_SPLIT_CHAINS"] = setting
    gen = api.SignalGenerator(prof, fl, k, mean, stdv, 42, num_workers=T, mode=api.MODE_CERTIFIED)
    gen.load_genome(contigs, 10000, api.SAMPLE_DNA)
    t0 = time.perf_counter()
    bs = [gen.sample(K) for _ in range(3)]
    t_stage = (time.perf_counter() - t0) / 3
    bs[0].run().wait()
    t0 = time.perf_counter()
    for b in bs[1:]:
        b.run()
    for b in bs[1:]:
        b.wait()
    dt = (time.perf_counter() - t0) / 2
    tm = gen.timing()
    print(f"-t {T} -K {K} {wl} split={'off' if setting else 'on '}: {bs[1].n_samples / dt:.3e} samples/s, {1e3 * dt:.2f} ms per batch "
          f"(events side {tm['events_ms'] + tm['dwell_ms']:.2f} ms, sample side {tm['samples_ms']:.2f} ms); staging {1e3 * t_stage:.1f} ms")
    for b in bs:
        b.free()
    gen.close()
