#!/usr/bin/env python3
"""Long-run self-consistency of the certified fp32 path: CERTIFIED vs EXACT mode, same seed, reads sampled on the device,
N bench-sized batches, every int16 compared (validate fast-vs-exact on >= 1e10 samples on the box).

    python tools/stress.py [n_batches] [profile]                  # T = K regime on the nCoV genome (host-side compare)
    python tools/stress.py --workload hg38-r10 [--samples 1.2e10]  # the headline workload: 3.09 Gb genome in HBM, -x dna-r10-prom,
                                                                  # -t 1, 16384 reads per batch; compared on the device
    python tools/stress.py --workload sequin-rna004 [--samples 1e10]

Prints one summary line (and, with --out FILE, appends it as a markdown table row: profiles/r03_soak.md keeps the runs)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from squigulator_amd import api, model, profiles  # noqa: E402


class _Dev:
    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def headline(args):
    import torch
    dev = torch.device("cuda", 0)
    torch.zeros(1, device=dev)
    wl_profile, wl_flags, wl_mode, wl_w, wl_k, _ = bench.WORKLOADS[args.workload]
    prof, fl = profiles.get_profile(args.profile or wl_profile)
    fl |= wl_flags
    k = profiles.default_kmer_size(fl)
    mean, stdv = model.synthetic_model(k)
    K = args.batch_reads or wl_k
    T = 1 if wl_w else K
    gens = [api.SignalGenerator(prof, fl, k, mean, stdv, 42, num_workers=T, mode=m) for m in (api.MODE_CERTIFIED, api.MODE_EXACT)]
    sm = api.SAMPLE_RNA if wl_mode == "rna" else api.SAMPLE_DNA
    if args.workload in ("hg38-r10", "synth-r10"):
        seq, lens = bench.synthetic_genome_device(args.genome_mb if args.genome_mb else (None if args.workload == "hg38-r10" else 64.0), dev)
        torch.cuda.synchronize()
        for g in gens:
            g.load_genome_device(seq.data_ptr(), lens, 10000, sm)
        del seq
        torch.cuda.empty_cache()
    else:
        contigs = bench.load_contigs(bench.SEQUINS if args.workload == "sequin-rna004" else bench.GENOME)
        for g in gens:
            g.load_genome(contigs, 10000, sm)
    workers = np.zeros(K, np.int32) if wl_w else None
    total = reads = fb = nb = 0
    t0 = time.time()
    while total < args.samples:
        bs = [g.sample(K, workers).run() for g in gens]
        for b in bs:
            b.wait()
        fb += gens[0].timing()["fallback_samples"]
        n = int(bs[0].n_samples)
        assert n == int(bs[1].n_samples) and np.array_equal(bs[0].sig_off, bs[1].sig_off), f"batch {nb}: lengths differ"
        assert bs[0].offset.tobytes() == bs[1].offset.tobytes() and bs[0].median_before.tobytes() == bs[1].median_before.tobytes()
        sig = [torch.as_tensor(_Dev(b.res.d_signal, n, "<i2"), device=dev) for b in bs]
        if not torch.equal(sig[0], sig[1]):
            bad = torch.nonzero(sig[0] != sig[1]).flatten()
            raise SystemExit(f"batch {nb}: {len(bad)} samples differ between certified and exact mode, first at {int(bad[0])}")
        ne = int(bs[0].n_events)
        dw = [torch.as_tensor(_Dev(b.res.d_dwell, ne, "<i2"), device=dev) for b in bs]      # (uint16 bit patterns)
        assert torch.equal(dw[0], dw[1]), f"batch {nb}: dwells differ"
        total += n; reads += bs[0].n_reads; nb += 1
        for b in bs:
            b.free()
    for g in gens:
        g.close()
    return (f"{args.workload} (-x {args.profile or wl_profile}, {'-t 1' if wl_w else 'T = K'}, {K} reads per batch)", nb, total, reads, fb, time.time() - t0)


def classic(nb, pname):
    prof, fl = profiles.get_profile(pname)
    k = profiles.default_kmer_size(fl)
    mean, stdv = model.synthetic_model(k)
    K = 8192
    genome = bench.load_genome(bench.GENOME)
    gens = [api.SignalGenerator(prof, fl, k, mean, stdv, 42, num_workers=K, mode=m) for m in (api.MODE_CERTIFIED, api.MODE_EXACT)]
    for g in gens:
        g.load_genome([genome], 10000, api.SAMPLE_DNA)
    bufs = None
    total = reads = fb = 0
    t0 = time.time()
    for it in range(nb):
        bs = [g.sample(K).run() for g in gens]
        for b in bs:
            b.wait()
        fb += gens[0].timing()["fallback_samples"]
        if bufs is None or len(bufs[0]) < bs[0].n_samples:
            bufs = [np.empty(int(bs[0].n_samples * 1.2), np.int16) for _ in gens]
        sigs = [b.signal(buf) for b, buf in zip(bs, bufs)]
        assert bs[0].n_samples == bs[1].n_samples and np.array_equal(bs[0].sig_off, bs[1].sig_off), f"batch {it}: lengths differ"
        assert np.array_equal(sigs[0], sigs[1]), f"batch {it}: signals differ"
        total += bs[0].n_samples; reads += bs[0].n_reads
        for b in bs:
            b.free()
    return (f"nCoV (-x {pname}, T = K, {K} reads per batch)", nb, total, reads, fb, time.time() - t0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("n_batches", nargs="?", type=int, default=40)
    ap.add_argument("profile_pos", nargs="?", default=None)
    ap.add_argument("--workload", default=None, choices=sorted(bench.WORKLOADS))
    ap.add_argument("--profile", default=None)
    ap.add_argument("--samples", type=float, default=1.2e10)
    ap.add_argument("--batch-reads", type=int, default=None)
    ap.add_argument("--genome-mb", type=float, default=None)
    ap.add_argument("--out", default=None, help="append the summary as a markdown table row to this file")
    args = ap.parse_args()
    what, nb, total, reads, fb, dt = headline(args) if args.workload else classic(args.n_batches, args.profile_pos or "dna-r9-prom")
    line = (f"{what}: {nb} batches, {reads} reads, {total:.4e} samples, certified == exact in every int16 (and every dwell); "
            f"{fb} samples ({fb / total:.2e}) took the FP64 fix-up; {dt:.0f} s")
    print(line)
    if args.out:
        new = not os.path.exists(args.out)
        with open(args.out, "a") as f:
            if new:
                f.write("# certified-vs-exact soak (tools/stress.py)\n\n| workload | batches | reads | samples compared | differing | FP64 fix-ups | wall s |\n|---|---|---|---|---|---|---|\n")
            f.write(f"| {what} | {nb} | {reads} | {total:.4e} | 0 | {fb} ({fb / total:.2e}) | {dt:.0f} |\n")


if __name__ == "__main__":
    main()
