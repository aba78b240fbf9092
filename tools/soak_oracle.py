#!/usr/bin/env python3
"""H3 soak: the HIP path against the CPU oracle (glibc libm) at scale -- SURVEY.md H3, round 3's review item 2.

The device evaluates FP64 `log` / `cos` / `sqrt` with ROCm's ocml, the reference with glibc (src/rand.h:87-94, src/gensig.c:264-270,
src/genread.c:202-205); each is < 1 ULP, they are not the same function.  The certified-vs-exact soaks (tools/stress.py) compare
device arithmetic with device arithmetic and say nothing about that.  This one runs the SAME job three times -- the oracle
(oracle/libsqg_oracle.so: the reference's arithmetic on the host's libm, one pthread per virtual worker), the HIP path in EXACT
mode (every sample through ocml's FP64) and in CERTIFIED mode (the fp32 path + FP64 fix-ups) -- and compares, per read: the sampled
read itself (contig, position, strand, length: `(int)grng`, `round(u*sum)`), `offset`, `median_before`, every dwell (`aln->ss`)
and every int16.

Regime: `-t T -K T*R` (T = the CPUs this process may use, so that the oracle's T workers run side by side; the reference's static
partition gives worker w the reads [w*R, (w+1)*R) of every batch, src/thread.c:80-99).  On the device that is the few-worker
path of the headline run (chains cut into links, bucketed hand-out for 9-mers) with T chains instead of one.

    python tools/soak_oracle.py [--profile dna-r10-prom] [--samples 1e11] [--genome-mb 64] [--out profiles/r04_soak.md]

A difference is reported with everything needed to replay it: batch, read, worker, event, sample index, both values (the stream
states are a function of (seed, worker, k-mer rank, samples drawn before), which the oracle can re-derive).  Test infrastructure:
only tools/ and tests/ load the oracle."""
import argparse
import ctypes as C
import os
import queue
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import orc  # noqa: E402
from squigulator_amd import api, model, profiles  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--profile", default="dna-r10-prom")
    ap.add_argument("--samples", type=float, default=1e11)
    ap.add_argument("--seconds", type=float, default=0.0, help="stop after this much wall time instead (0: by --samples)")
    ap.add_argument("--genome-mb", type=float, default=64.0)
    ap.add_argument("--workers", type=int, default=0, help="T (default: the CPUs this process may use)")
    ap.add_argument("--reads-per-worker", type=int, default=128)
    ap.add_argument("--modes", default="exact,certified")
    ap.add_argument("--out", default=None, help="append the summary as a markdown table row to this file")
    args = ap.parse_args()

    aff, quota = bench.cpu_allowance()
    T = args.workers or max(1, min(aff, int(quota) if quota else aff))
    K = T * args.reads_per_worker
    prof, fl = profiles.get_profile(args.profile)
    rna = bool(fl & profiles.SQ_RNA)
    if rna:
        fl |= profiles.SQ_PREFIX
    k = profiles.default_kmer_size(fl)
    mean, stdv = model.synthetic_model(k)
    contigs = bench.load_contigs(bench.SEQUINS) if rna else bench.synthetic_genome_host(args.genome_mb)
    modes = [(m, api.MODE_EXACT if m == "exact" else api.MODE_CERTIFIED) for m in args.modes.split(",")]

    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as tmp:
        fa = os.path.join(tmp, "g.fa")
        with open(fa, "wb") as f:
            for i, c in enumerate(contigs):
                f.write(b">c%d\n" % i + c + b"\n")
        orac = orc.Oracle(prof, fl, k, mean, stdv, 42, num_workers=T)
        orac.load_ref(fa)
    gens = [api.SignalGenerator(prof, fl, k, mean, stdv, 42, num_workers=T, mode=m) for _, m in modes]
    for g in gens:
        g.load_genome(contigs, 10000, api.SAMPLE_RNA if rna else api.SAMPLE_DNA)

    # the oracle runs ahead in a thread of its own (ctypes drops the GIL; its T pthreads are the box's cores); the main thread
    # queues the device batches, fetches them and compares
    L = orac.L
    q = queue.Queue(maxsize=2)
    stop = threading.Event()

    def producer():
        while not stop.is_set():
            q.put(L.orc_batch_run(orac.core, orac.ref, K, 1, T))
        q.put(None)

    th = threading.Thread(target=producer, daemon=True)
    t0 = time.time()
    th.start()
    total = reads = events = nb = 0
    fixups = 0
    diffs = []
    sigbuf = None
    # one batch staged ahead of the one being queued, as a streaming host does: every sqg_batch_run finds its successor staged, so that the
    # successor's first event pass rides along with this batch's hand-out (k_part_hand_count) -- the path the bench times
    nxt = [g.sample(K).run() for g in gens]
    ahead = [g.sample(K) for g in gens]
    while True:
        cur, nxt = nxt, ahead
        ahead = [g.sample(K) for g in gens]
        for b in nxt:
            b.run()
        ob = q.get()
        rd = ob.contents.reads
        for (mname, _), g, b in zip(modes, gens, cur):
            b.wait()
            if mname == "certified":
                fixups += max(g.timing()["fallback_samples"], 0)
            if sigbuf is None or len(sigbuf) < b.n_samples:
                sigbuf = np.empty(int(b.n_samples * 1.3), np.int16)
            sig = b.signal(sigbuf)
            dw = b.dwell()
            smp = b.sampled
            for i in range(K):
                r = rd[i]
                n = int(r.len_raw_signal)
                where = f"batch {nb} read {i} (worker {r.tid}, {mname})"
                if (smp["ref_idx"][i], smp["ref_pos"][i], smp["rlen"][i], smp["strand"][i:i + 1]) != (r.ref_idx, r.ref_pos_st, r.rlen, r.strand):
                    diffs.append(f"{where}: sampled read differs: device (contig {smp['ref_idx'][i]}, pos {smp['ref_pos'][i]}, len {smp['rlen'][i]}, "
                                 f"{smp['strand'][i:i + 1]}) oracle ({r.ref_idx}, {r.ref_pos_st}, {r.rlen}, {r.strand})")
                    continue
                if b.offset[i] != r.offset or b.median_before[i] != r.median_before:
                    diffs.append(f"{where}: offset / median_before differ: {b.offset[i]!r} / {b.median_before[i]!r} vs {r.offset!r} / {r.median_before!r}")
                e0, e1 = int(b.ev_off[i]), int(b.ev_off[i + 1])
                oss = np.ctypeslib.as_array(r.ss, shape=(int(r.ss_n),))
                if e1 - e0 != r.ss_n or not np.array_equal(dw[e0:e1], oss):
                    j = int(np.argmax(dw[e0:e0 + len(oss)] != oss)) if e1 - e0 == r.ss_n else -1
                    diffs.append(f"{where}: dwell differs at event {j}: device {int(dw[e0 + j]) if j >= 0 else e1 - e0} oracle {int(oss[j]) if j >= 0 else r.ss_n} "
                                 "(time stream: rand.h:87-94 through round(), gensig.c:254-257)")
                    continue
                s0, s1 = int(b.sig_off[i]), int(b.sig_off[i + 1])
                osig = np.ctypeslib.as_array(r.raw_signal, shape=(n,))
                if s1 - s0 != n or not np.array_equal(sig[s0:s1], osig):
                    bad = np.nonzero(sig[s0:s0 + n] != osig)[0] if s1 - s0 == n else np.zeros(0, np.int64)
                    for j in bad[:8]:
                        gi = int(j) if not rna else n - 1 - int(j)           # generation index (RNA signals are stored reversed)
                        ev = int(np.searchsorted(np.cumsum(oss), gi, side="right"))
                        diffs.append(f"{where}: sample {int(j)} (event {ev}, draw {gi - int(np.cumsum(oss)[ev - 1] if ev else 0)} of it): device {int(sig[s0 + j])} "
                                     f"oracle {int(osig[j])} (gensig.c:264-270: nrng -> float -> s*dig/range-offset -> int16)")
                    if s1 - s0 != n:
                        diffs.append(f"{where}: {s1 - s0} samples on the device, {n} in the oracle")
            b.free()
        total += sum(int(rd[i].len_raw_signal) for i in range(K))
        events += sum(int(rd[i].ss_n) for i in range(K))
        reads += K; nb += 1
        L.orc_batch_free(ob)
        if len(diffs) > 200 or (args.seconds > 0 and time.time() - t0 > args.seconds) or (args.seconds <= 0 and total >= args.samples):
            break
    stop.set()
    while q.get() is not None:
        pass
    for b in nxt:
        b.wait(); b.free()
    for b in ahead:
        b.free()
    for g in gens:
        g.close()
    orac.close()
    dt = time.time() - t0
    what = (f"`-x {args.profile}{' --prefix=yes' if rna else ''}` `-t {T} -K {K}`, "
            f"{'sequins' if rna else f'{args.genome_mb:.0f} Mb synthetic genome'}; HIP {' + '.join(m for m, _ in modes)} vs oracle (glibc)")
    line = (f"{what}: {nb} batches, {reads} reads, {events:.4e} dwells, {total:.4e} samples per mode compared with the oracle; "
            f"{len(diffs)} differences; {fixups} FP64 fix-ups in certified mode; {dt:.0f} s")
    print(line)
    for d in diffs[:50]:
        print("  DIFF", d)
    if args.out:
        new = not os.path.exists(args.out)
        with open(args.out, "a") as f:
            if new:
                f.write("# HIP path vs CPU oracle (glibc libm) soak -- tools/soak_oracle.py\n\n"
                        "Every sampled read (contig, position, strand, length), `offset`, `median_before`, dwell and int16 of the device's output "
                        "compared with the oracle's on the GPU box's host, per mode.\n\n"
                        "| job | batches | reads | dwells compared | samples compared (per mode) | differing | FP64 fix-ups (certified) | wall s |\n|---|---|---|---|---|---|---|---|\n")
            f.write(f"| {what} | {nb} | {reads} | {events:.4e} | {total:.4e} | {len(diffs)} | {fixups} ({fixups / max(total, 1):.2e}) | {dt:.0f} |\n")
            for d in diffs[:50]:
                f.write(f"\n* DIFF {d}\n")
    sys.exit(1 if diffs else 0)


if __name__ == "__main__":
    main()
