#!/usr/bin/env python3
"""bench.py -- simulated raw samples/s of the per-read signal path on N MI355X.

Workload (BASELINE.json configs[1]): nCoV-2019 reference, -x dna-r9-prom (R9 6-mer pore model),
reads of gamma-distributed length (-r 10000) cut from the 29 903-nt genome, --seed 42.  One "step" is
one batch (one process_db()) of --batch-reads reads in the T=K regime: read i of a batch runs on
virtual worker i.  Steps x batch-reads reads in total; the default 12 x 32768 = 4 x the config's n=100000.
The reads are the ones the reference itself would draw: gen_read (src/genread.c) with `--seed 42 -t T -K T`
on the genome kept in HBM, sampled by the library's device-side sampler at staging time (--host-sampler:
numpy draws of the same distribution, uploaded).  Inputs (sequences, per-read descriptors) are resident in
HBM before the timed region starts.

N>1: one process per GPU (torch.distributed, backend nccl = RCCL).  The job's T = N*K virtual workers
are sharded contiguously over ranks; each rank stages and runs only its own workers' reads (no
steady-state collective; one broadcast of the pore model at start-up) => weak scaling.

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel (k_samples_lean) against HBM:
achieved = algorithmic bytes (2*N_samples + N_bases + 24*N_reads per launch) / its
average launch duration measured with hipEvents on the library's stream.  `cpu_baseline` is the
oracle (a C restatement of the reference's path, oracle/) timed on this box's host cores on a bounded
sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from squigulator_amd import api, model, profiles, shard  # noqa: E402

HBM_PEAK_BYTES_PER_S = 8.0e12   # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md
GENOME = os.path.join(ROOT, "tests", "golden", "inputs", "nCoV-2019.reference.fasta")


def load_genome(path):
    seq = []
    with open(path) as f:
        for line in f:
            if not line.startswith(">"):
                seq.append(line.strip())
    return "".join(seq).encode()


_COMP = bytes.maketrans(b"ACGTacgt", b"TGCATGCA")


def sample_reads(genome: bytes, n: int, rlen: int, rng: np.random.Generator):
    """Reads with the reference sampler's distribution (src/genread.c:243-281): length ~ Erlang-2 with
    scale rlen/2, start uniform over the genome, clipped at the contig end, >=200 nt, random strand."""
    out = []
    G = len(genome)
    while len(out) < n:
        m = n - len(out)
        lens = rng.gamma(2.0, rlen / 2, size=m).astype(np.int64)
        pos = rng.integers(0, G, size=m)
        strand = rng.integers(0, 2, size=m)
        for L, p, s in zip(lens, pos, strand):
            r = genome[p:p + L]
            if len(r) < 200:
                continue
            out.append(r if s else r.translate(_COMP)[::-1])
    return out[:n]


HG38_CHR_MB = [248, 242, 198, 190, 182, 171, 159, 145, 138, 134, 135, 133, 114, 107, 102, 90, 83, 80, 59, 64, 47, 51, 156, 57]


def load_contigs(path):
    out, cur = [], []
    with open(path) as f:
        for line in f:
            if line.startswith(">"):
                if cur:
                    out.append("".join(cur).encode())
                cur = []
            else:
                cur.append(line.strip())
    if cur:
        out.append("".join(cur).encode())
    return out


def synthetic_genome(total_mb: float, seed: int = 1):
    """hg38 is not available offline: 24 contigs with hg38's chromosome proportions, i.i.d. uniform ACGT
    from a fixed seed, a few N runs (telomere/centromere-like) to exercise the sampler's rejection rule."""
    rng = np.random.default_rng(seed)
    lut = np.frombuffer(b"ACGT", np.uint8)
    scale = total_mb * 1e6 / (sum(HG38_CHR_MB) * 1e6)
    contigs = []
    for mb in HG38_CHR_MB:
        n = max(int(mb * 1e6 * scale), 5000)
        a = lut[rng.integers(0, 4, n, dtype=np.uint8)]
        a[:min(1000, n // 50)] = ord("N")
        mid = n // 3
        a[mid:mid + min(3000, n // 20)] = ord("N")
        contigs.append(a.tobytes())
    return contigs


WORKLOADS = {
    # name: (profile, extra flags, sampler mode, description)
    "ncov-r9": ("dna-r9-prom", 0, "dna", "nCoV-2019.reference.fasta -x dna-r9-prom (BASELINE.json configs[1])"),
    "synth-r10": ("dna-r10-prom", 0, "dna", "synthetic hg38-proportioned genome -x dna-r10-prom (configs[2]/[3]; hg38 itself is not available offline)"),
    "sequin-rna004": ("rna004-prom", profiles.SQ_PREFIX, "rna", "rnasequin_sequences_2.4.fa -x rna004-prom --prefix=yes, whole transcripts (configs[4])"),
}


def pack(reads):
    off = np.zeros(len(reads) + 1, np.int64)
    off[1:] = np.cumsum([len(r) for r in reads])
    return b"".join(reads), off


def pmc_traffic(profile, batch_reads, rlen, mode):
    """HBM bytes per k_samples launch from the rocprofv3 PMC passes of the same command
    (tools/prof_pmc.sh -> profiles/traffic_latest.json); None when no profile of this workload exists.
    The counters cannot be read from inside the process being timed, so the bench quotes the committed
    measurement of the identical configuration."""
    path = os.path.join(ROOT, "profiles", "traffic_latest.json")
    try:
        with open(path) as f:
            doc = json.load(f)
        if doc.get("workload_key") == f"{profile}|batch_reads={batch_reads}|rlen={rlen}|mode={mode}":
            return float(doc["kernels"]["k_samples_lean"]["hbm_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        pass
    return None


def cpu_baseline(prof, flags, k, mean, stdv, genome, rlen, target_cpu_seconds=20.0, check_mode=None):
    """Oracle (oracle/libsqg_oracle.so) on the host cores, T=K regime, bounded sample.  With check_mode the same
    reads also go through the HIP path and every int16 is compared (the oracle as the checker): `parity`."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes
    import orc
    cores = os.cpu_count() or 1
    # the oracle (like the reference) mallocs a >128 KiB signal buffer per read; with hundreds of threads glibc's
    # default mmap threshold turns that into mmap/munmap storms.  Keep those allocations on the heap so the
    # baseline measures generation, not the kernel's mmap lock.
    try:
        libc = ctypes.CDLL("libc.so.6")
        libc.mallopt(-3, 1 << 30)      # M_MMAP_THRESHOLD
        libc.mallopt(-1, 1 << 30)      # M_TRIM_THRESHOLD
    except OSError:
        pass
    rng = np.random.default_rng(1234)
    # calibrate the single-core rate on a few reads, then size the timed sample for ~target_cpu_seconds
    # of CPU work in total (and at least 4 reads per core so every core has work)
    probe = sample_reads(genome, 8, rlen, rng)
    o = orc.Oracle(prof, flags, k, mean, stdv, 42, num_workers=len(probe))
    t0 = time.perf_counter()
    res = o.run_batch_seqs(probe, want_ss=False, nthreads=1)
    dt = time.perf_counter() - t0
    o.close()
    ns = sum(len(r.sig) for r in res)
    rate1 = ns / dt                                   # one core, samples/s
    mean_len = ns / len(probe)
    n = int(min(max(target_cpu_seconds * rate1 / mean_len, 4 * cores), 20000))
    reads = sample_reads(genome, n, rlen, rng)
    o = orc.Oracle(prof, flags, k, mean, stdv, 42, num_workers=n)
    t0 = time.perf_counter()
    res = o.run_batch_seqs(reads, want_ss=False, nthreads=cores)
    dt = time.perf_counter() - t0
    o.close()
    ns = sum(len(r.sig) for r in res)
    out = {"value": ns / dt, "unit": "samples/s", "cores": cores, "kind": "port",
           "sample": f"{n} reads / {ns} samples of the same workload, generation only (no BLOW5 encode), "
                     f"-t {n} -K {n} on {cores} host threads, {dt:.2f} s wall"}
    if check_mode is not None:
        # the very same reads, seed and workers through the C ABI: bit-for-bit comparison of the whole sample
        gen = api.SignalGenerator(prof, flags, k, mean, stdv, seed=42, num_workers=n, mode=check_mode)
        b = gen.submit(reads)
        sig = b.signal()
        bad = 0
        for i, r in enumerate(res):
            got = sig[b.sig_off[i]:b.sig_off[i + 1]]
            if len(got) != len(r.sig) or not np.array_equal(got, r.sig):
                bad += 1
        out["parity"] = {"reads": n, "samples": int(ns), "reads_differing": bad, "equal": bad == 0 and int(b.n_samples) == int(ns)}
        b.free(); gen.close()
    return out


def cpu_reference(prof, flags, k, mean, stdv, nproc, reads_per_proc=24, rlen=10000):
    """The REFERENCE's own gensig.c/genread.c (oracle/_ref/ref_harness, compiled in the build container from the
    upstream sources where they lie) timed on the host cores: `nproc` independent single-threaded processes
    (-t1, different seeds), aggregate samples/s.  Includes the reference's read sampling and the harness's
    binary dump; returns None when the binary did not travel."""
    import subprocess
    import tempfile
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    if not os.path.exists(harness):
        return None
    with tempfile.TemporaryDirectory() as tmp:
        mpath = os.path.join(tmp, "m.model")
        model.write_f5c_model(mpath, k, mean, stdv)
        cfgs = []
        for i in range(nproc):
            cfg = {"fasta": GENOME, "model": mpath, "out": os.path.join(tmp, f"o{i}.bin"), "flags": flags, "amp_noise": 1.0,
                   "seed": 1000 + 7919 * i, "threads": 1, "batch": 1000, "nreads": reads_per_proc, "rlen": rlen}
            for name, v in zip(("digitisation", "sample_rate", "bps", "range", "offset_mean", "offset_std",
                                "median_before_mean", "median_before_std", "dwell_mean", "dwell_std"), prof.as_tuple()):
                cfg[name] = repr(float(v))
            cp = os.path.join(tmp, f"c{i}.txt")
            with open(cp, "w") as f:
                f.write("".join(f"{a}={b}\n" for a, b in cfg.items()))
            cfgs.append(cp)
        t0 = time.perf_counter()
        procs = [subprocess.Popen([harness, c], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for c in cfgs]
        rcs = [p.wait() for p in procs]
        dt = time.perf_counter() - t0
        if any(rcs):
            return None
        import struct
        ns = 0
        for i in range(nproc):
            # dump layout (oracle/ref_harness.c): "SQGREF1\0", int32 n; per read int32[6] (…, rlen at [4], …),
            # f64 offset, f64 median, i64 len_raw_signal, i64 start_time, i64 ss_n, seq, int16 signal, int32 ss
            with open(os.path.join(tmp, f"o{i}.bin"), "rb") as f:
                f.seek(8)
                (n,) = struct.unpack("<i", f.read(4))
                for _ in range(n):
                    hdr = struct.unpack("<6i", f.read(24))
                    _, _, ln, _, ssn = struct.unpack("<ddqqq", f.read(40))
                    ns += ln
                    f.seek(hdr[4] + 2 * ln + 4 * ssn, 1)
    return {"value": ns / dt, "unit": "samples/s", "cores": nproc, "kind": "reference",
            "sample": f"{nproc} x ref_harness -t1 -n {reads_per_proc} (reference gensig.c/genread.c, synthetic table), "
                      f"{dt:.2f} s wall incl. process start, model/FASTA load and the dump"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch-reads", type=int, default=None,
                    help="reads per step per GPU (= virtual workers per GPU).  Default 32768 (6-mer: 1.4e11 samples/s at 512, "
                         "4.8e11 at 8192, 5.4e11 at 32768 and 65536); 16384 for the 9-mer workload, whose 1-MiB-per-worker "
                         "state table makes larger batches slower (1.2e11 at 1024, 1.63e11 at 8192-16384, 1.54e11 at 32768)")
    ap.add_argument("--rlen", type=int, default=10000)
    ap.add_argument("--workload", default="ncov-r9", choices=sorted(WORKLOADS),
                    help="ncov-r9 is the headline (BASELINE.json configs[1]); the others are the remaining configs, reported for information")
    ap.add_argument("--genome-mb", type=float, default=64.0, help="size of the synthetic genome of --workload synth-r10")
    ap.add_argument("--profile", default=None, help="override the workload's -x preset")
    ap.add_argument("--mode", default="certified", choices=["exact", "certified"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-sampler", action="store_true",
                    help="sample the reads with numpy on the host (same distribution) instead of the device-side gen_read")
    ap.add_argument("--no-store-probe", action="store_true")
    ap.add_argument("--job-workers", type=int, default=0,
                    help="T of the whole job (the reference's -t).  Default: one worker per read, sharded by worker over the "
                         "GPUs, no data-path collective.  With a value (e.g. 1: the reference's reproducible regime) every "
                         "GPU owns all T workers and generates a range of each batch's reads; the per-stream sample counts "
                         "are all-gathered over RCCL once per batch (range sharding, include/sqg.h)")
    ap.add_argument("--workers-per-gpu", type=int, default=0,
                    help="W > 0: the job has T = N*W virtual workers, W per GPU (sharded by worker, no data-path collective), and "
                         "every batch of N*K reads is split over them as the reference's static partition does "
                         "(src/thread.c:80-99): `-t N*W -K N*K`.  W = 1 on one GPU is the reference's reproducible `-t 1`")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL) for one rank per GPU; gloo only to exercise the N>1 control flow on a "
                         "box with fewer GPUs than ranks (ranks then share GPUs)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the signal path has no CPU fallback")
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and world > ndev:
        raise SystemExit(f"{world} ranks but {ndev} GPUs (use --backend gloo to share GPUs in a dry run)")
    local_rank %= ndev
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    wl_profile, wl_flags, wl_mode, wl_desc = WORKLOADS[args.workload]
    if args.batch_reads is None:
        args.batch_reads = 16384 if args.workload == "synth-r10" else 32768
    if args.profile is None:
        args.profile = wl_profile
    prof, flags = profiles.get_profile(args.profile)
    flags |= wl_flags
    k = profiles.default_kmer_size(flags)
    n_k = 1 << (2 * k)
    # pore model: rank 0 owns it; RCCL broadcast to the other GPUs (the only collective on this path)
    if rank == 0:
        mean, stdv = model.synthetic_model(k)
    else:
        mean, stdv = np.zeros(n_k, np.float32), np.zeros(n_k, np.float32)
    if world > 1:
        mean, stdv = shard.broadcast_model(mean, stdv, src=0)

    K = args.batch_reads
    range_mode = args.job_workers > 0
    if range_mode and args.host_sampler:
        raise SystemExit("--job-workers needs the device sampler")
    W = args.workers_per_gpu
    if W and (range_mode or K % W):
        raise SystemExit("--workers-per-gpu: not with --job-workers, and it must divide --batch-reads")
    T = args.job_workers if range_mode else (W * world if W else K * world)
    w_lo, w_hi = (0, T) if range_mode else shard.worker_range(rank, world, T)   # default: contiguous block of K virtual workers
    gen = api.SignalGenerator(prof, flags, k, mean, stdv, seed=42, num_workers=T, device=local_rank,
                              mode=api.MODE_EXACT if args.mode == "exact" else api.MODE_CERTIFIED,
                              worker_lo=w_lo, worker_hi=w_hi)
    if range_mode:
        gen.set_range_mode(True)
    r_lo, r_hi = shard.read_range(rank, world, K * world)          # range mode: my reads of the job's K*world-read batches
    if args.workload == "synth-r10":
        contigs = synthetic_genome(args.genome_mb)
    elif args.workload == "sequin-rna004":
        contigs = load_contigs(os.path.join(ROOT, "tests", "golden", "inputs", "rnasequin_sequences_2.4.fa"))
    else:
        contigs = [load_genome(GENOME)]
    genome = contigs[0]
    rng = np.random.default_rng(42 + rank)
    # worker of this GPU's read i: one each (T = K), or the static partition of the job's N*K-read batch over T = N*W workers
    workers = np.arange(w_lo, w_hi, dtype=np.int32) if not W else (w_lo + np.arange(K, dtype=np.int32) // (K // W)).astype(np.int32)

    nsteps = args.warmup + args.steps
    batches = []
    if args.host_sampler:
        for _ in range(nsteps):
            blob, off = pack(sample_reads(genome, K, args.rlen, rng))
            batches.append(gen.stage_packed(blob, off, workers))  # H2D happens here, outside the timed region
    else:
        # the reads ARE the reference's: gen_read (src/genread.c) with `--seed 42 -r <rlen> -t T -K T` on the
        # resident genome, sampled on the device at staging time (outside the timed region)
        gen.load_genome(contigs, args.rlen, api.SAMPLE_RNA if wl_mode == "rna" else api.SAMPLE_DNA)
        for _ in range(nsteps):
            batches.append(gen.sample(K * world, None, lo=r_lo, hi=r_hi) if range_mode else gen.sample(K, workers))

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    n_rows = T * n_k
    keep = []

    def run(b):
        """default: one asynchronous launch sequence.  Range mode: counts -> all-gather in range order -> the rest"""
        if not range_mode or not gen_has_streams:
            return b.run()
        mine = shard.counts_tensor(b.run_begin(), n_rows, torch.device("cuda", local_rank))
        if world > 1:
            before, after = shard.exchange_counts(mine)
        else:
            before = after = torch.zeros_like(mine)
        torch.cuda.synchronize()
        keep.append((before, after))                               # alive until the batch has run
        return b.run_end(before.data_ptr(), after.data_ptr())

    gen_has_streams = not (flags & (profiles.SQ_IDEAL | profiles.SQ_IDEAL_AMP))
    iso_lean_ms = []                  # warm-up batches run one at a time: the sample kernel alone on the machine
    for b in batches[:args.warmup]:
        run(b).wait()
        iso_lean_ms.append(gen.timing()["lean_ms"])
    sync_all()
    t0 = time.perf_counter()
    sig_ms, dwell_ms, ev_ms, lean_ms = [], [], [], []
    samples = bases = reads = 0
    for b in batches[args.warmup:]:
        run(b)                        # asynchronous: all K steps are queued back to back (range mode: one exchange per step)
    for b in batches[args.warmup:]:
        b.wait()
        tm = gen.timing()
        sig_ms.append(tm["samples_ms"]); dwell_ms.append(tm["dwell_ms"]); ev_ms.append(tm["events_ms"])
        lean_ms.append(tm["lean_ms"] if tm["lean_ms"] > 0 else tm["samples_ms"])
        samples += b.n_samples; bases += b.n_bases; reads += b.n_reads
    sync_all()
    dt = time.perf_counter() - t0

    tot = torch.tensor([float(samples), float(reads), dt], dtype=torch.float64,
                       device="cuda" if (world == 1 or args.backend == "nccl") else "cpu")
    if world > 1:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dt_max = float(mx[2])
    else:
        dt_max = dt
    tot_samples, tot_reads = float(tot[0]), float(tot[1])

    if rank == 0:
        steps = max(args.steps, 1)
        alg_bytes = (2 * samples + bases + 24 * reads) / steps           # per k_samples_lean launch (this rank)
        k_ms = float(np.mean(lean_ms)) if lean_ms else float("nan")     # the dominant kernel alone
        achieved = alg_bytes / (k_ms * 1e-3)
        out = {
            "metric": "simulated raw samples/sec",
            "value": tot_samples / dt_max,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_max / steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64" if args.mode == "exact" else "f32+f64",
            "data": "synthetic",
            "config": {
                "workload": f"{wl_desc}; -x {args.profile} --seed 42 -r {args.rlen}, "
                            + (f"-t {T} -K {K * world} (range sharding: every GPU owns all {T} worker(s) and generates {K} reads of each "
                               f"batch; one all-gather of {4 * n_rows} B per batch), {args.steps} batches" if range_mode else
                               f"-t {T} -K {K * world} ({W} worker(s) per GPU, sharded by worker; static partition of each batch, "
                               f"{K} reads per GPU), {args.steps} batches" if W else
                               f"-t {T} -K {T} (T=K virtual workers, {K} per GPU), {args.steps} batches"),
                "reads_per_step_per_gpu": K, "kmer_size": k, "mode": args.mode,
                "reads": "numpy draws of gen_read's distribution (host)" if args.host_sampler
                         else "gen_read on the device-resident genome (library sampler), as the reference with these options",
                "pore_model": "synthetic stand-in table (built-in ONT tables absent from the reference mount)",
            },
            "reads_per_s": tot_reads / dt_max,
            "samples_per_step_per_gpu": samples / steps,
            "kernel_ms": {"k_samples_lean": k_ms, # from the end of k_events to the batch's last kernel; the fix-ups run on their own stream next to the NEXT batch's
                          # k_events and are stretched by it, so this span is longer than the kernels in it
                          "k_scan..k_fixup* (span; fix-ups overlap the next k_events)": float(np.mean(sig_ms)) if sig_ms else None,
                          "k_events(+dwell)": float(np.mean(ev_ms)) if ev_ms else None,
                          "k_dwell(separate)": float(np.mean(dwell_ms)) if dwell_ms and np.mean(dwell_ms) > 0.01 else None},
            "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK_BYTES_PER_S / 1e9,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_BYTES_PER_S,
                         "traffic": pmc_traffic(args.profile, K, args.rlen, args.mode),
                         # with SQG_OVERLAP=1 the next batch's k_events runs next to this kernel and stretches it; the
                         # warm-up batches run one at a time and give the kernel's duration alone on the machine
                         "kernel_ms": k_ms,
                         "kernel_ms_warmup": (iso_lean_ms[-1] if iso_lean_ms and iso_lean_ms[-1] > 0 else None),
                         "streams": 2 if os.environ.get("SQG_OVERLAP") else 1,
                         "kernel": "k_samples_lean" if args.mode == "certified" else "k_samples<exact>", "algorithmic_bytes_per_launch": alg_bytes},
        }
        if not args.no_store_probe:
            out["roofline"]["measured_store_peak_GBps"] = gen.probe_store_bandwidth(1 << 30, 10) / 1e9
        if not args.no_cpu_baseline and (args.workload != "ncov-r9" or world > 1):
            out["cpu_baseline"] = None                       # the CPU legs: headline workload, single-GPU run only
        elif not args.no_cpu_baseline:
            # the reference's own gensig.c/genread.c (oracle/_ref, kind "reference") when the harness travelled with the
            # repo, else the oracle restatement (kind "port"); the other one is reported next to it
            port = cpu_baseline(prof, flags, k, mean, stdv, genome, args.rlen,
                                check_mode=api.MODE_EXACT if args.mode == "exact" else api.MODE_CERTIFIED)
            out["parity_check"] = port.pop("parity", None)
            ref = cpu_reference(prof, flags, k, mean, stdv, nproc=min(os.cpu_count() or 1, 128), reads_per_proc=150)
            out["cpu_baseline"] = ref or port
            if ref:
                out["cpu_port"] = port
        print(json.dumps(out))
    for b in batches:
        b.free()
    gen.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
