#!/usr/bin/env python3
"""bench.py -- simulated raw samples/s of the per-read signal path on N MI355X.

Headline workload (BASELINE.json configs[2], the configuration the metric is quoted on): a synthetic hg38-scale
genome -- 24 contigs with hg38's chromosome lengths (3.09 Gb, i.i.d. ACGT, telomere / acrocentric / heterochromatin-like
N runs: hg38 itself is not available offline) kept in HBM -- `-x dna-r10-prom` (R10 9-mer pore model),
`-r 10000 --seed 42`, reads drawn by the library's device-side gen_read (src/genread.c).  One "step" is one batch
(one process_db()) of --batch-reads reads per GPU.  Regime: ONE virtual worker per GPU (`-t N -K N*batch`, the
reference's static partition, src/thread.c:80-99); on one GPU that is the reference's reproducible `-t 1`.  A worker's
reads of a batch form one chain; the library cuts it into links and, for 9-mers, hands the k-mer streams out over events
bucketed by the top bits of the rank (squigulator_amd/csrc/k_part.h).  Inputs (sequences, descriptors) are resident in HBM
before the timed region starts; the K timed steps are queued back to back.

Other workloads (--workload): ncov-r9 (configs[1], T=K regime), sequin-rna004 (configs[4]), synth-r10 (a small genome).

N>1: one process per GPU (torch.distributed, backend nccl = RCCL).  Started without WORLD_SIZE, `--gpus N` launches the N
ranks itself (python -m torch.distributed.run on 127.0.0.1).  The job's T virtual workers are sharded contiguously over
the ranks; each rank stages and runs only its own workers' reads (no steady-state collective; one broadcast of the pore
model at start-up) => weak scaling.  --job-workers T: range sharding instead (every rank owns all T workers and
generates a range of each batch; one all-gather of per-stream sample counts per batch).

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel (k_samples_lean) against HBM: achieved =
algorithmic bytes (2*N_samples + N_bases + 24*N_reads per launch) / its average launch duration measured
with hipEvents on the library's stream; `step_frac` is the same bytes over the whole step, `step_traffic(_frac)` all the HBM bytes of a
step (PMC, every kernel of the timed region) over the step's time.  `cpu_baseline` is the
reference's own gensig.c/genread.c (oracle/_ref/ref_harness) timed on this box's host cores, one `-t 1` process per
physical core, read loop only (kind "reference"); the oracle restatement (kind "port") when that binary is absent.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import shutil

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # (multi-process GPU work on these hosts: dmabuf IPC only; must be set before HIP starts)
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from squigulator_amd import api, model, profiles, shard  # noqa: E402

HBM_PEAK_BYTES_PER_S = 8.0e12   # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md
GENOME = os.path.join(ROOT, "tests", "golden", "inputs", "nCoV-2019.reference.fasta")
SEQUINS = os.path.join(ROOT, "tests", "golden", "inputs", "rnasequin_sequences_2.4.fa")

# GRCh38 primary assembly, chr1..22, X, Y
HG38_LEN = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422,
            135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167,
            46709983, 50818468, 156040895, 57227415]
# N runs besides 10 kb at both ends of every contig: (contig, start as a fraction of its length, length as a fraction) --
# the acrocentric short arms, the large heterochromatin gaps of chr1/9/16 and of chrY: 4.9 % of the genome, as in hg38
HG38_NRUNS = [(12, 0.0, 0.14), (13, 0.0, 0.15), (14, 0.0, 0.167), (20, 0.0, 0.107), (21, 0.0, 0.207),
              (0, 0.49, 0.072), (8, 0.31, 0.13), (15, 0.40, 0.10), (23, 0.47, 0.52)]


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


def genome_layout(total_mb: float | None):
    """contig lengths (hg38's, or scaled to total_mb) and the N runs as (absolute start, length)"""
    scale = 1.0 if total_mb is None else total_mb * 1e6 / sum(HG38_LEN)
    lens = [max(int(n * scale), 5000) for n in HG38_LEN]
    off = np.concatenate([[0], np.cumsum(lens)])
    tel = max(int(10000 * min(scale * 30, 1.0)), 50)
    runs = []
    for c, n in enumerate(lens):
        runs.append((int(off[c]), min(tel, n // 20)))
        runs.append((int(off[c]) + n - min(tel, n // 20), min(tel, n // 20)))
    for c, s, l in HG38_NRUNS:
        runs.append((int(off[c]) + int(s * lens[c]), int(l * lens[c])))
    return lens, runs


def synthetic_genome_host(total_mb: float, seed: int = 1):
    """the same layout on the host (numpy), for the CPU legs and the tests: list of contigs (bytes)"""
    lens, runs = genome_layout(total_mb)
    rng = np.random.default_rng(seed)
    a = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, sum(lens), dtype=np.uint8)]
    for s, l in runs:
        a[s:s + l] = ord("N")
    off = np.concatenate([[0], np.cumsum(lens)])
    return [a[off[i]:off[i + 1]].tobytes() for i in range(len(lens))]


def synthetic_genome_device(total_mb: float | None, device, seed: int = 1):
    """the synthetic genome made in HBM (torch is the allocator here): (uint8 tensor of the concatenated contigs, lengths)"""
    import torch
    lens, runs = genome_layout(total_mb)
    total = sum(lens)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    seq = torch.empty(total, dtype=torch.uint8, device=device)
    step = 1 << 28
    for s in range(0, total, step):
        c = torch.randint(0, 4, (min(step, total - s),), dtype=torch.uint8, device=device, generator=g)
        # 0,1,2,3 -> 'A','C','G','T' = 65,67,71,84
        seq[s:s + len(c)] = 65 + 2 * c + 2 * (c >= 2).to(torch.uint8) + 11 * (c == 3).to(torch.uint8)
    for s, l in runs:
        seq[s:s + l] = ord("N")
    return seq, lens


_COMP = bytes.maketrans(b"ACGTacgt", b"TGCATGCA")


def sample_reads_host(contigs, n: int, rlen: int, rng: np.random.Generator):
    """Reads with the reference sampler's distribution (src/genread.c:243-281) drawn with numpy: length ~ Erlang-2 with
    scale rlen/2, start uniform over the genome, clipped at the contig end, >= 200 nt, <= 10 % N, random strand."""
    out = []
    cum = np.cumsum([len(c) for c in contigs])
    while len(out) < n:
        m = n - len(out)
        lens = rng.gamma(2.0, rlen / 2, size=m).astype(np.int64)
        pos = rng.integers(0, cum[-1], size=m)
        strand = rng.integers(0, 2, size=m)
        for L, p, s in zip(lens, pos, strand):
            ci = int(np.searchsorted(cum, p, side="right"))
            q = int(p - (cum[ci - 1] if ci else 0))
            r = contigs[ci][q:q + int(L)]
            if len(r) < 200 or r.count(b"N") * 10 > len(r):
                continue
            r = r.replace(b"N", b"A")
            out.append(r if s else r.translate(_COMP)[::-1])
    return out[:n]


def load_genome(path):
    """a single-contig FASTA as bytes"""
    return b"".join(load_contigs(path))


def sample_reads(genome: bytes, n: int, rlen: int, rng: np.random.Generator):
    """sample_reads_host on one contig (tests and tools)"""
    return sample_reads_host([genome], n, rlen, rng)


def pack(reads):
    off = np.zeros(len(reads) + 1, np.int64)
    off[1:] = np.cumsum([len(r) for r in reads])
    return b"".join(reads), off


WORKLOADS = {
    # name: (profile, extra flags, sampler mode, workers per GPU (0: one per read), reads per step per GPU, description)
    "hg38-r10": ("dna-r10-prom", 0, "dna", 1, 32768,
                 "synthetic hg38-scale genome (24 contigs, hg38's chromosome lengths, 3.09 Gb) -x dna-r10-prom (BASELINE.json configs[2]; "
                 "configs[3] at N=8)"),
    "ncov-r9": ("dna-r9-prom", 0, "dna", 0, 32768, "nCoV-2019.reference.fasta -x dna-r9-prom (BASELINE.json configs[1])"),
    "synth-r10": ("dna-r10-prom", 0, "dna", 1, 8192, "small synthetic hg38-proportioned genome (--genome-mb) -x dna-r10-prom"),
    "sequin-rna004": ("rna004-prom", profiles.SQ_PREFIX, "rna", 1, 32768,
                      "rnasequin_sequences_2.4.fa -x rna004-prom --prefix=yes, whole transcripts (configs[4])"),
}


STEP_KERNELS = ("k_part_events", "k_part_tile_bases", "k_part_offsets", "k_part_slices", "k_part_slice_bounds", "k_part_hist", "k_part_scan",
                "k_part_hand", "k_events", "k_link_prefix", "k_scan", "k_items", "k_samples", "k_fixup")


def _traffic_doc(workload_key):
    """profiles/traffic_latest.json if it belongs to THIS workload and to THESE kernels: the file is stamped with a hash of the
    library's sources (squigulator_amd.build.source_hash) by tools/make_traffic.py; counters of other kernels are not quoted"""
    from squigulator_amd import build as _b
    path = os.path.join(ROOT, "profiles", "traffic_latest.json")
    try:
        with open(path) as f:
            doc = json.load(f)
        if doc.get("workload_key") != workload_key or doc.get("source_hash") != _b.source_hash():
            return None
        return doc
    except (OSError, KeyError, ValueError):
        return None


def pmc_step_traffic(workload_key):
    """HBM bytes of one whole step -- every kernel of the timed region, one launch each -- from the same PMC passes
    (profiles/traffic_latest.json); None when no profile of this workload and this build exists"""
    doc = _traffic_doc(workload_key)
    try:
        if doc is None:
            return None
        ks = doc["kernels"]
        # with the next batch staged ahead (the timed region: everything is) a step's hand-out and the next step's counting pass are ONE
        # launch (k_part_hand_count); the two kernels it replaces only run for the first batch and are not part of a steady-state step
        return step_traffic_of(ks)
    except (KeyError, ValueError):
        return None


def pmc_traffic(workload_key):
    """HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes of the same command
    (tools/prof_pmc.sh -> profiles/traffic_latest.json); None when no profile of this workload and this build exists.
    The counters cannot be read from inside the process being timed: the default run measures them in two child passes behind its
    own legs (live_traffic); this is the committed measurement of the identical configuration and sources, quoted when that did not run."""
    doc = _traffic_doc(workload_key)
    try:
        return None if doc is None else float(doc["kernels"]["k_samples_lean"]["hbm_bytes_per_launch"])
    except (KeyError, ValueError):
        return None


def resources_from(kk, cal, kernel_ms, samples_per_launch, store_peak_GBps, source):
    """What the dominant kernel occupies besides HBM bytes, from its PMC counters per launch (`kk`: SQ_ACTIVE_INST_VALU, SQ_INSTS_VALU,
    GRBM_GUI_ACTIVE, TCP_TCC_READ/WRITE_REQ_sum, FETCH/WRITE_SIZE_KiB, hbm_bytes_per_launch, kernel_us of the GRBM pass) priced with THIS
    run's kernel time: `valu_busy` = SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs (quad-cycles -> cycles);
    `l2_req_frac` = the kernel's TCP -> L2 requests per cycle over what `cal_rgather8` (tools/pmc_calib.hip: random 8-B look-ups in an
    L2-resident table, `cal` requests per cycle) sustains; `store_frac` = written bytes per second over the box's int16 streaming-store
    rate; `hbm_frac` = all HBM bytes per second over 8 TB/s.  `bound` names the largest."""
    try:
        if not kernel_ms or "SQ_ACTIVE_INST_VALU" not in kk or "GRBM_GUI_ACTIVE" not in kk:
            return None
        cyc = kk["GRBM_GUI_ACTIVE"] / 8.0
        out = {"source": source,
               "valu_busy": kk["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / cyc,
               "valu_inst_per_sample": kk["SQ_INSTS_VALU"] * 64.0 / samples_per_launch if samples_per_launch else None,
               "kernel_cycles_pmc": cyc, "kernel_clock_GHz_pmc": cyc / (kk.get("kernel_us", 0) * 1e3) if kk.get("kernel_us") else None}
        # the kernel at single issue: one 4-cycle pass per VALU wave-instruction and SIMD, a second pass for each of the loop's three transcendentals
        # per 64 samples (gfx950 pairs VALU instructions only in streams without transcendentals: profiles/r04_ubench.md) -- at the clock the
        # counters were taken at.  Next to kernel_ms it says how much of the kernel's time is anything BUT its instruction count
        if out["kernel_clock_GHz_pmc"] and samples_per_launch:
            cyc_issue = (kk["SQ_INSTS_VALU"] * 4.0 + 3.0 * (samples_per_launch / 64.0) * 4.0) / 1024.0
            out["valu_single_issue_ms"] = cyc_issue / (out["kernel_clock_GHz_pmc"] * 1e9) * 1e3
            out["kernel_ms_over_valu_single_issue"] = kernel_ms / out["valu_single_issue_ms"]
        if "TCP_TCC_READ_REQ_sum" in kk or "TCP_TCC_WRITE_REQ_sum" in kk:
            req = kk.get("TCP_TCC_READ_REQ_sum", 0.0) + kk.get("TCP_TCC_WRITE_REQ_sum", 0.0)
            out["l2_req_per_sample"] = req / samples_per_launch if samples_per_launch else None
            out["l2_req_per_cycle"] = req / cyc
            out["l2_req_frac"] = (req / cyc / cal) if cal else None
        else:
            out["l2_req_per_sample"] = out["l2_req_per_cycle"] = out["l2_req_frac"] = None
        out["l2_req_calibrated_peak_per_cycle"] = cal
        wbytes = kk["WRITE_SIZE_KiB"] * 1024.0
        out["store_GBps"] = wbytes / (kernel_ms * 1e-3) / 1e9
        out["store_frac"] = (out["store_GBps"] / store_peak_GBps) if store_peak_GBps else None
        out["hbm_frac"] = kk["hbm_bytes_per_launch"] / (kernel_ms * 1e-3) / HBM_PEAK_BYTES_PER_S
        cands = {"valu": out["valu_busy"], "l2_requests": out["l2_req_frac"], "stores": out["store_frac"], "hbm": out["hbm_frac"]}
        cands = {k: v for k, v in cands.items() if v is not None}
        out["bound"] = max(cands, key=cands.get)
        # what `valu_busy` is and is not: cycles in which the SIMDs hold a VALU instruction in flight, over the launch's cycles.  It names the
        # largest share, not a proven limiter: a build of the same kernel with 21 % fewer VALU instructions per sample (two samples per lane,
        # -DSQG_LEAN_PAIR=1 without its last phase) takes the same time (profiles/r06_pair.md) -- the memory pipeline (the signal stream at
        # the box's store rate + one pore-table row and one stream state per event) stands right behind it
        out["bound_caveat"] = ("largest share of the four, not a proven limiter: 21 % fewer VALU instructions per sample leave the kernel's time "
                               "unchanged (profiles/r06_pair.md); the memory pipeline -- stores at the box's rate + per-event gathers -- is right behind")
        return out
    except (KeyError, ValueError, ZeroDivisionError, TypeError):
        return None


def stored_calibration():
    """requests per cycle `cal_rgather8` sustained when profiles/traffic_latest.json was measured: a property of the chip, not of the
    library's sources (quoted whatever the file's hash)"""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic_latest.json")) as f:
            return (json.load(f).get("calib") or {}).get("rgather8_l2_req_per_cycle")
    except (OSError, ValueError):
        return None


def pmc_resources(workload_key, kernel_ms, samples_per_launch, store_peak_GBps):
    """resources_from() on the hash-matched PMC passes (tools/prof_pmc.sh -> profiles/traffic_latest.json); None without a matching
    profile"""
    doc = _traffic_doc(workload_key)
    try:
        if doc is None:
            return None
        return resources_from(doc["kernels"]["k_samples_lean"], (doc.get("calib") or {}).get("rgather8_l2_req_per_cycle"), kernel_ms,
                              samples_per_launch, store_peak_GBps,
                              "profiles/traffic_latest.json (PMC passes of the same sources and workload), priced with this run's kernel_ms")
    except (KeyError, ValueError, TypeError):
        return None


def _pmc_short(kn):
    """kernel name of a rocprofv3 CSV row -> the key profiles/*_traffic.json uses (template arguments kept for the event passes only)"""
    short = kn.split("(")[0].split("<")[0].replace("void ", "").strip()
    if short in ("k_events", "k_part_events"):
        short = kn.split("(")[0].replace("void ", "").strip()
    return short


def parse_pmc_csv(path, counter):
    """{kernel: [value per launch]} of ONE counter from a rocprofv3 `*_counter_collection.csv` (a launch's row per counter; rocprofv3
    sums the counter's instances)"""
    import csv
    acc = {}
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") == counter:
                acc.setdefault(_pmc_short(r["Kernel_Name"]), []).append(float(r["Counter_Value"]))
    return acc


def traffic_from_pmc(fetch_kib, write_kib):
    """HBM bytes per launch and kernel from the two passes' per-launch lists: FETCH_SIZE and WRITE_SIZE are KiB, FETCH_SIZE is doubled
    (the gfx950 correction of MI355X_MICROARCH.md's HBM section, as tools/make_traffic.py); a kernel needs both counters"""
    out = {}
    for k in set(fetch_kib) & set(write_kib):
        f = sum(fetch_kib[k]) / len(fetch_kib[k])
        w = sum(write_kib[k]) / len(write_kib[k])
        out[k] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0,
                  "launches": min(len(fetch_kib[k]), len(write_kib[k]))}
    return out


def step_traffic_of(ks):
    """HBM bytes of one steady-state step from per-kernel averages: every kernel of the timed region, one launch each.  With the next
    batch staged ahead (the timed region: everything is) a step's hand-out and the next step's counting pass are ONE launch
    (k_part_hand_count); the two kernels it replaces only run for a run's first batch and are not part of a steady-state step"""
    fused = any(k.startswith("k_part_hand_count") for k in ks)
    skip = ("k_part_hand_ord", "k_part_events<1, 0>", "k_part_events<2, 0>") if fused else ()
    return float(sum(v["hbm_bytes_per_launch"] for k, v in ks.items() if k.startswith(STEP_KERNELS) and not k.startswith(skip)))


def under_profiler():
    """is this process already a profiler's child (rocprofv3 preloads its tool library and exports ROCP_* / ROCPROF*)?"""
    return (any(k.startswith(("ROCP_", "ROCPROF", "ROCPROFILER")) for k in os.environ)
            or "rocprof" in os.environ.get("LD_PRELOAD", ""))


LIVE_PASSES = (            # (name, counters of ONE rocprofv3 --pmc pass, extra arguments of the child): combinations tools/prof_pmc.sh has run all round
    ("fetch", ("FETCH_SIZE",), ("--no-store-probe",)),
    ("write", ("WRITE_SIZE",), ()),                                              # (with the store probe: 1 GiB per launch checks the unit)
    ("valu", ("GRBM_GUI_ACTIVE", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU"), ("--no-store-probe",)),
    ("l2req", ("TCP_TCC_READ_REQ_sum", "TCP_TCC_WRITE_REQ_sum"), ("--no-store-probe",)),
)


def live_traffic(args, limit_s=180.0):
    """roofline.traffic and roofline.resources as measurements of THIS box and THESE libraries: `rocprofv3 --pmc` passes (counters only;
    FETCH_SIZE and WRITE_SIZE each alone, as MI355X_MICROARCH.md prescribes; then the VALU issue counters with the launch's cycles, then the
    TCP -> L2 requests) of this script in its shortest form (3 timed steps behind one warm-up step of the same workload, no other leg), each
    a child process behind this run's own legs, ~2.5 s apiece.  The WRITE_SIZE pass keeps the store probe: k_store_probe writes exactly
    1 GiB per launch, which checks the counter's unit.  If tools/bin/pmc_calib is there (__graft_entry__.build() compiles it), its random
    8-B look-ups are counted too: the L2 request rate this chip sustains.  None when rocprofv3 is not there, the run is itself profiled, or
    one of the two traffic passes fails or exceeds the time limit (the line then quotes the hash-matched profile, as before); a failing
    later pass only leaves its counters out."""
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None or under_profiler():
        return None
    child = [sys.executable, os.path.abspath(__file__), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--pipeline-seconds", "0",
             "--e2e-seconds", "0", "--small-batch-seconds", "0", "--every-batch-launches", "0", "--live-traffic", "off",
             "--workload", args.workload, "--rlen", str(args.rlen), "--mode", args.mode]
    if args.batch_reads is not None:
        child += ["--batch-reads", str(args.batch_reads)]
    if args.genome_mb is not None:
        child += ["--genome-mb", str(args.genome_mb)]
    if args.profile is not None:
        child += ["--profile", args.profile]
    if args.order_free:
        child += ["--order-free"]
    if args.lib is not None:
        child += ["--lib", args.lib]
    if args.workers_per_gpu is not None:
        child += ["--workers-per-gpu", str(args.workers_per_gpu)]
    t0 = time.perf_counter()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    got, dur, calib, failed = {}, {}, None, []

    def one_pass(tmp, name, counters, cmd_tail):
        left = limit_s - (time.perf_counter() - t0)
        if left < 10:
            return None
        cmd = [exe, "--pmc"] + list(counters) + ["--kernel-trace", "--output-format", "csv", "-d", tmp, "-o", name, "--"] + cmd_tail
        # (a process group of its own: a pass that exceeds its limit is ended with its children -- the profiled python -- not just the profiler)
        with subprocess.Popen(cmd, cwd=tmp, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True) as pr:
            try:
                _, err_txt = pr.communicate(timeout=left)
            except subprocess.TimeoutExpired:
                import signal
                try:
                    os.killpg(pr.pid, signal.SIGKILL)
                except OSError:
                    pass
                pr.communicate()
                raise
        p = subprocess.CompletedProcess(cmd, pr.returncode, "", err_txt)
        path = os.path.join(tmp, name + "_counter_collection.csv")
        if p.returncode != 0 or not os.path.exists(path):
            print(f"[bench] live counters: the {name} pass failed (rc {p.returncode}): {p.stderr[-300:]}", file=sys.stderr)
            return None
        return path

    try:
        with tempfile.TemporaryDirectory(dir="/tmp" if os.path.isdir("/tmp") else None) as tmp:
            env["TMPDIR"] = tmp
            for name, counters, extra in LIVE_PASSES:
                try:
                    path = one_pass(tmp, name, counters, child + list(extra))
                except subprocess.SubprocessError as e:
                    print(f"[bench] live counters: the {name} pass: {type(e).__name__}", file=sys.stderr)
                    path = None
                if path is None:
                    if name in ("fetch", "write"):
                        return None
                    failed.append(name)
                    continue
                for c in counters:
                    got[c] = parse_pmc_csv(path, c)
                if name == "valu":                                   # the launch durations of the pass whose cycles are quoted: the clock
                    import csv
                    kt = os.path.join(tmp, name + "_kernel_trace.csv")
                    if os.path.exists(kt):
                        with open(kt, newline="") as f:
                            for r in csv.DictReader(f):
                                dur.setdefault(_pmc_short(r["Kernel_Name"]), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            cal_bin = os.path.join(ROOT, "tools", "bin", "pmc_calib")
            if os.access(cal_bin, os.X_OK) and "l2req" not in failed:
                try:
                    path = one_pass(tmp, "cal", ("TCP_TCC_READ_REQ_sum", "GRBM_GUI_ACTIVE"), [cal_bin, "2"])
                    if path is not None:
                        rq = parse_pmc_csv(path, "TCP_TCC_READ_REQ_sum"); cy = parse_pmc_csv(path, "GRBM_GUI_ACTIVE")
                        k = next((k for k in rq if "cal_rgather8" in k), None)
                        if k and cy.get(k):
                            calib = (sum(rq[k]) / len(rq[k])) / (sum(cy[k]) / len(cy[k]) / 8.0)
                except subprocess.SubprocessError:
                    pass
    except (OSError, ValueError, KeyError) as e:
        print(f"[bench] live counters: {type(e).__name__}: {e}", file=sys.stderr)
        return None
    ks = traffic_from_pmc(got["FETCH_SIZE"], got["WRITE_SIZE"])
    for c in ("GRBM_GUI_ACTIVE", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "TCP_TCC_READ_REQ_sum", "TCP_TCC_WRITE_REQ_sum"):
        for k, v in got.get(c, {}).items():
            if k in ks and v:
                ks[k][c] = sum(v) / len(v)
    for k, v in dur.items():
        if k in ks and v:
            ks[k]["kernel_us"] = sum(v) / len(v)
    if "k_samples_lean" not in ks and "k_samples" not in ks:
        return None
    probe = got["WRITE_SIZE"].get("k_store_probe")
    return {"kernels": ks, "seconds": time.perf_counter() - t0, "calib_rgather8_l2_req_per_cycle": calib, "failed_passes": failed,
            # 1.0 = WRITE_SIZE counts KiB (k_store_probe writes 2^30 bytes per launch)
            "write_size_unit_check": None if not probe else (sum(probe) / len(probe)) / float(1 << 20),
            "what": "rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE; GRBM_GUI_ACTIVE + SQ_ACTIVE_INST_VALU + SQ_INSTS_VALU; TCP_TCC_READ/WRITE_REQ_sum) of "
                    "`bench.py --steps 3 --warmup 1` on this box, this workload and this library, run behind the line's own legs; per launch, "
                    "averaged over the pass' launches; FETCH_SIZE doubled (gfx950)"}


def host_info():
    """(CPU model, physical cores, hardware threads)"""
    name, phys = "unknown", set()
    try:
        with open("/proc/cpuinfo") as f:
            pid = cid = None
            for line in f:
                if line.startswith("model name") and name == "unknown":
                    name = line.split(":", 1)[1].strip()
                elif line.startswith("physical id"):
                    pid = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    cid = line.split(":")[1].strip()
                elif not line.strip():
                    if pid is not None and cid is not None:
                        phys.add((pid, cid))
                    pid = cid = None
    except OSError:
        pass
    threads = os.cpu_count() or 1
    return name, (len(phys) or threads), threads


def cpu_allowance():
    """CPUs this process may actually use: (scheduler affinity, cgroup quota in CPUs or None).  A container that shows every
    core of the host but is throttled to a few would make an N-process baseline look N/few times slower per core."""
    try:
        aff = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        aff = os.cpu_count() or 1
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                t = f.read().split()
            if path.endswith("cpu.max"):
                if t and t[0] != "max":
                    quota = float(t[0]) / float(t[1])
            else:
                q = float(t[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                        quota = q / float(g.read().split()[0])
            break
        except (OSError, ValueError, IndexError):
            continue
    return aff, quota


def _cpulist(text):
    out = []
    for part in text.strip().split(","):
        if part:
            a, _, b = part.partition("-")
            out += list(range(int(a), int(b or a) + 1))
    return out


def pin_to_gpu_numa_node(device_index, local_rank, local_world):
    """One process per GPU: keep this rank's host threads (the staging helpers, zlib) on the CPUs of the NUMA node its GPU hangs off,
    and -- ranks that share a node -- on a slice of their own.  Best effort: returns what was done, or why nothing was."""
    import torch
    try:
        p = torch.cuda.get_device_properties(device_index)
        bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as f:
            node = int(f.read())
        if node < 0:
            return {"pinned": False, "why": "the GPU reports no NUMA node", "pci": bdf}
        with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
            cpus = sorted(set(_cpulist(f.read())) & set(os.sched_getaffinity(0)))
        if not cpus:
            return {"pinned": False, "why": "no usable CPU on the GPU's node", "pci": bdf, "numa_node": node}
        # the ranks whose GPUs share this node (same answer on every one of them: the devices are enumerated alike)
        mates = []
        for d in range(min(local_world, torch.cuda.device_count())):
            q = torch.cuda.get_device_properties(d)
            try:
                with open(f"/sys/bus/pci/devices/{q.pci_domain_id:04x}:{q.pci_bus_id:02x}:{q.pci_device_id:02x}.0/numa_node") as f:
                    if int(f.read()) == node:
                        mates.append(d)
            except (OSError, ValueError):
                pass
        if device_index in mates and len(mates) > 1 and len(cpus) >= len(mates):
            i, m = mates.index(device_index), len(mates)
            cpus = cpus[len(cpus) * i // m:len(cpus) * (i + 1) // m]
        os.sched_setaffinity(0, cpus)
        return {"pinned": True, "pci": bdf, "numa_node": node, "cpus": len(cpus), "first_cpu": cpus[0]}
    except (OSError, ValueError, AttributeError, RuntimeError) as e:
        return {"pinned": False, "why": f"{type(e).__name__}: {e}"}


def cpu_reference(prof, flags, k, mean, stdv, fasta_path, rlen, seconds=10.0, full_fasta=None):
    """The REFERENCE's own gensig.c/genread.c (oracle/_ref/ref_harness, compiled in the build container from the upstream
    sources where they lie), timed mode: every process loads the model and the FASTA, seeds its streams, then generates
    reads for `seconds` of wall time and reports the samples and the time of that read loop alone.
    Legs: T=1 alone on the machine; one -t 1 process per physical core (generation only); the same writing BLOW5 through
    slow5lib (zlib + svb-zd, as the reference does).  Returns None when the binary did not travel."""
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    if not os.path.exists(harness):
        return None
    cpu, phys_cores, threads = host_info()
    aff, quota = cpu_allowance()
    cores = max(1, min(phys_cores, aff, int(quota) if quota else phys_cores))     # processes that really get a core each
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as tmp:
        mpath = os.path.join(tmp, "m.model")
        model.write_f5c_model(mpath, k, mean, stdv)

        def leg(nproc, secs, blow5, fasta=None):
            cfgs = []
            for i in range(nproc):
                cfg = {"fasta": fasta or fasta_path, "model": mpath, "flags": flags & ~profiles.SQ_ORDER_FREE, "amp_noise": 1.0, "seed": 1000 + 7919 * i,
                       "threads": 1, "batch": 1000, "nreads": 1, "rlen": rlen, "time_s": secs}
                if blow5:
                    cfg["slow5"] = os.path.join(tmp, f"o{i}.blow5")
                for name, v in zip(("digitisation", "sample_rate", "bps", "range", "offset_mean", "offset_std",
                                    "median_before_mean", "median_before_std", "dwell_mean", "dwell_std"), prof.as_tuple()):
                    cfg[name] = repr(float(v))
                cp = os.path.join(tmp, f"c{i}.txt")
                with open(cp, "w") as f:
                    f.write("".join(f"{a}={b}\n" for a, b in cfg.items()))
                cfgs.append(cp)
            procs = [subprocess.Popen([harness, c], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for c in cfgs]
            outs = [p.communicate()[0] for p in procs]
            if any(p.returncode for p in procs):
                return None
            ns = rd = 0
            tmax = 0.0
            for o in outs:
                kv = dict(t.split("=") for t in o.split() if "=" in t)
                ns += int(kv["samples"]); rd += int(kv["reads"]); tmax = max(tmax, float(kv["loop_seconds"]))
            for i in range(nproc):
                try:
                    os.unlink(os.path.join(tmp, f"o{i}.blow5"))
                except OSError:
                    pass
            return ns / tmax, rd / tmax, tmax

        one = leg(1, min(seconds, 5.0), False)
        allc = leg(cores, seconds, False)
        e2e = leg(cores, seconds, True)
        # ... and ONE process on the headline genome itself (3.09 Gb: the sampler's look-ups leave the caches; the per-sample work is the same)
        full = leg(1, min(seconds, 5.0), False, fasta=full_fasta) if full_fasta else None
    if not one or not allc:
        return None
    return {"value": allc[0], "unit": "samples/s", "cores": cores, "kind": "reference",
            "per_core": allc[0] / cores, "t1": one[0], "reads_per_s": allc[1],
            "to_blow5": (e2e[0] if e2e else None), "t1_full_genome": (full[0] if full else None),
            "full_genome_bases": (os.path.getsize(full_fasta) if full_fasta else None), "cpu": cpu, "physical_cores": phys_cores, "hw_threads": threads,
            "affinity_cpus": aff, "cgroup_cpu_quota": quota,
            "sample": f"reference gensig.c/genread.c (oracle/_ref/ref_harness, gcc -O2 -std=c99 as the reference's Makefile), synthetic "
                      f"pore table, same profile and -r on a {os.path.getsize(fasta_path) / 1e6:.0f} MB genome of the same layout; `value`: "
                      f"{cores} x `-t 1` processes (one per core this container may use), {allc[2]:.1f} s of read loop each, generation only (no "
                      f"output); `t1`: one process alone; `to_blow5`: the same {cores} processes writing BLOW5 (zlib+svb-zd) through slow5lib; `t1_full_genome`: one process "
                      f"alone on the timed run's own genome (the same bytes, copied from HBM to a FASTA in /dev/shm)"}


def cpu_port(prof, flags, k, mean, stdv, batches, T, nthreads):
    """the oracle restatement (oracle/libsqg_oracle.so) on the same batches of reads, one after the other (the streams carry over)
    -> (per-batch per-read results, samples/s)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    o = orc.Oracle(prof, flags, k, mean, stdv, 42, num_workers=T)
    t0 = time.perf_counter()
    res = [o.run_batch_seqs(reads, want_ss=False, nthreads=nthreads if T > 1 else 1) for reads in batches]      # (the reference's static partition over T)
    dt = time.perf_counter() - t0
    o.close()
    return res, sum(len(r.sig) for rb in res for r in rb) / dt


def parity_check(prof, flags, k, mean, stdv, contigs, rlen, n_reads, one_worker, mode):
    """the oracle as the checker: n_reads host-drawn reads of the workload, in three batches, through the oracle and through the C ABI in
    the bench's regime -- one worker: the chains are cut into links, bucketed hand-out for 9-mers; else one worker per read -- and in the
    timed region's launch pattern: every batch is run with its successor already staged (whose first event pass then rides along with
    this batch's hand-out, k_part_hand_count).  Every int16 compared."""
    rng = np.random.default_rng(1234)
    reads = sample_reads_host(contigs, n_reads, rlen, rng)
    per = (n_reads + 2) // 3
    batches = [reads[i:i + per] for i in range(0, n_reads, per)]
    T = 1 if one_worker else max(len(bt) for bt in batches)
    res, rate = cpu_port(prof, flags & ~profiles.SQ_ORDER_FREE, k, mean, stdv, batches, T, nthreads=min(os.cpu_count() or 1, 64))
    gen = api.SignalGenerator(prof, flags, k, mean, stdv, seed=42, num_workers=T, mode=mode)
    bad = ns_gpu = 0

    def compare(b, want):
        nonlocal bad, ns_gpu
        sig = b.signal()
        ns_gpu += int(b.n_samples)
        for i, r in enumerate(want):
            got = sig[b.sig_off[i]:b.sig_off[i + 1]]
            if len(got) != len(r.sig) or not np.array_equal(got, r.sig):
                bad += 1
        b.free()
    staged = [gen.stage(batches[0], np.zeros(len(batches[0]), np.int32) if one_worker else None)]
    for i in range(len(batches)):
        if i + 1 < len(batches):
            staged.append(gen.stage(batches[i + 1], np.zeros(len(batches[i + 1]), np.int32) if one_worker else None))
        staged[i].run()
        if i >= 1:
            compare(staged[i - 1].wait(), res[i - 1])
    compare(staged[-1].wait(), res[-1])
    ns = int(sum(len(r.sig) for rb in res for r in rb))
    out = {"reads": n_reads, "samples": ns, "reads_differing": bad, "equal": bad == 0 and ns_gpu == ns,
           "regime": "-t 1" if one_worker else "T = K", "batches": len(batches), "staged_ahead": True}
    gen.close()
    return out, rate


def parity_check_rank(prof, flags, k, mean, stdv, contigs, rlen, T, w_lo, w_hi, per_worker, mode, device):
    """N > 1: THIS rank's shard against the oracle's run of the WHOLE job, so that the first real multi-GPU run certifies itself.
    A small job of the bench's regime -- T virtual workers, two batches of per_worker * T host-drawn reads, the reference's static partition
    (src/thread.c:84-101) -- goes through the oracle with all T workers on every rank (the same seeded reads everywhere); a fresh context that owns
    the workers [w_lo, w_hi) of T, as this rank's bench context does, generates their reads; every int16 compared, two batches (carried state).
    Returns [reads differing, reads, samples, a 62-bit digest of the rank's signals]."""
    rng = np.random.default_rng(4321)
    n_b = per_worker * T
    batches = [sample_reads_host(contigs, n_b, rlen, rng) for _ in range(2)]
    res, _ = cpu_port(prof, flags & ~profiles.SQ_ORDER_FREE, k, mean, stdv, batches, T, nthreads=max(1, min(4, (os.cpu_count() or 1))))
    gen = api.SignalGenerator(prof, flags, k, mean, stdv, seed=42, num_workers=T, device=device, mode=mode, worker_lo=w_lo, worker_hi=w_hi)
    bad = n = ns = 0
    h = hashlib.sha256()
    for reads, want in zip(batches, res):
        mine = [i for i, r in enumerate(want) if w_lo <= r.tid < w_hi]
        b = gen.stage([reads[i] for i in mine], np.array([want[i].tid for i in mine], np.int32)).run().wait()
        sig = b.signal()
        for j, i in enumerate(mine):
            got = sig[b.sig_off[j]:b.sig_off[j + 1]]
            if len(got) != len(want[i].sig) or not np.array_equal(got, want[i].sig):
                bad += 1
        h.update(sig.tobytes())
        n += len(mine); ns += int(b.n_samples)
        b.free()
    gen.close()
    return [bad, n, ns, int.from_bytes(h.digest()[:8], "little") >> 2]


def pipeline_leg(stage_one, run, n_batches, seconds=None, max_batches=40000):
    """Nothing staged ahead but the batch behind the one being queued: ONE host thread samples + stages batch i+2 (device-side gen_read,
    descriptors, links), queues batch i+1 -- the library lets the first event pass of the staged batch behind it ride along with its
    hand-out (include/sqg.h, sqg_batch_run) --, waits for batch i and frees it.  Two batches in flight, one staged, results left in HBM.
    With `seconds` the leg runs on the clock -- at least n_batches, then until `seconds` have elapsed (at most max_batches) -- and the batch
    count it did is the fifth element; without, exactly n_batches (range sharding has a collective per batch: every rank the same count).
    Returns (samples, reads, seconds, host seconds spent in the sampler + staging call per batch, batches run)."""
    cur = run(stage_one())
    nxt = stage_one()
    samples = reads = 0
    t_stage = 0.0
    t0 = time.perf_counter()
    done = 0
    while done < n_batches or (seconds is not None and done < max_batches and time.perf_counter() - t0 < seconds):
        a = time.perf_counter()
        nn = stage_one()
        t_stage += time.perf_counter() - a
        run(nxt)
        cur.wait()
        samples += cur.n_samples; reads += cur.n_reads
        cur.free()
        cur, nxt = nxt, nn
        done += 1
    cur.wait()
    samples += cur.n_samples; reads += cur.n_reads
    dt = time.perf_counter() - t0
    cur.free()
    nxt.free()                                                   # (staged, never run)
    return samples, reads, dt, t_stage / max(done, 1), done + 1


def e2e_legs(gen, prof, flags, sample_batch, reads_per_batch, seconds, kinds=("pinned_int16", "pinned_svb", "blow5", "blow5_fast")):
    """The other half of work_per_single_read / output_db (src/sim.c:602-611,630-641): what a host that drains the results gets
    (SURVEY.md 8d / H5).  Three legs, one host thread each, batch i+1 sampled + staged + queued before batch i is consumed:
    `pinned_int16` -- sqg_fetch_signal into sqg_host_alloc memory (raw int16 over PCIe); `pinned_svb` -- sqg_batch_compress (svb-zd on
    the device, the signal field of a BLOW5 record), then batch i+1 queued, then sqg_fetch_svb; `blow5` -- sqg_blow5_write_batch into /dev/shm (record framing +
    zlib on the host's threads: the bytes the reference writes); `blow5_fast` -- the same call on a writer opened with SQG_BLOW5_STORED (records
    framed on the device in stored-block zlib streams: a valid BLOW5 file with the reference's records, not its bytes); `blow5_fast_4files` --
    the same on four files (SQG_BLOW5_SHARDS(4): one file of a tmpfs takes 6.9 GB/s whatever the writers, four take four times that).  Never `value`: PCIe and zlib are 20x and 1500x below the kernels."""
    import torch
    probe = sample_batch().run().wait()
    cap = int(probe.n_samples * 1.4) + 65536
    probe.free()
    pin16 = gen.pinned(2 * cap, np.int16)
    pin8 = gen.pinned(3 * cap, np.uint8)                   # (svb-zd: 1.0-1.3 B per sample on these signals; 3.25 at worst, checked by the binding)
    ids = [b"S1_%d!c0!0!10000!+" % i for i in range(reads_per_batch)]
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    out = {"reads_per_batch": reads_per_batch,
           "what": "one host thread: sample + stage + queue batch i+1, wait for batch i, drain it; samples/s of the drained batches"}
    for kind in kinds:
        path = os.path.join(shm, f"sqg_bench_e2e_{os.getpid()}.blow5")
        # blow5: the reference's bytes (deflate on the host's threads); blow5_fast: SQG_BLOW5_STORED -- the same records in stored-block
        # zlib streams, framed on the device, one PCIe copy and one pwrite() behind the next batch (include/sqg.h)
        w = api.Blow5Writer(path, prof, flags, threads=0, stored=kind.startswith("blow5_fast"), shards=4 if kind == "blow5_fast_4files" else 1) if kind.startswith("blow5") else None
        w_paths = list(w.paths) if w is not None else [path]
        samples = nb = nbytes = 0

        def drain(b):
            nonlocal nbytes
            if kind == "pinned_int16":
                b.signal(out=pin16); nbytes += 2 * b.n_samples
            elif kind == "pinned_svb":
                nbytes += len(b.fetch_svb(out=pin8))
            else:
                w.write_batch(b, ids)
        try:
            cur = sample_batch().run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            while True:
                last = nb >= 1 and time.perf_counter() - t0 >= seconds
                if kind in ("pinned_svb", "blow5_fast", "blow5_fast_4files"):       # the encoder first (the context has ONE buffer of encodings: it cannot run ahead),
                    cur.compress(fetch=False)                  # the next batch's kernels behind it -- they run while the bytes cross PCIe
                nxt = None if last else sample_batch().run()
                cur.wait()
                drain(cur)
                samples += cur.n_samples; nb += 1
                cur.free()
                if nxt is None:
                    break
                cur = nxt
            if w is not None:
                nbytes = w.close(); w = None
            dt = time.perf_counter() - t0
        finally:
            if w is not None:
                w.close()
            for q in w_paths:
                try:
                    os.unlink(q)
                except OSError:
                    pass
        out[kind] = {"value": samples / dt, "unit": "samples/s", "seconds": dt, "batches": nb, "bytes_per_sample": nbytes / max(samples, 1),
                     "reads_per_batch": reads_per_batch, "GBps": nbytes / dt / 1e9}
    return out


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100,
                    help="timed steps (default 100: a timed region of ~0.36 s at the headline size; 20 steps, 75 ms, were the same size as the box-to-box spread)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="hg38-r10", choices=sorted(WORKLOADS))
    ap.add_argument("--batch-reads", type=int, default=None, help="reads per step per GPU (default: per workload)")
    ap.add_argument("--rlen", type=int, default=10000)
    ap.add_argument("--genome-mb", type=float, default=None,
                    help="hg38-r10 / synth-r10: scale the synthetic genome to this many Mb (default: hg38's 3088 Mb; synth-r10: 64)")
    ap.add_argument("--profile", default=None, help="override the workload's -x preset")
    ap.add_argument("--mode", default="certified", choices=["exact", "certified"])
    ap.add_argument("--order-free", action="store_true",
                    help="SQG_ORDER_FREE in cfg.flags: the few-worker stream hand-out by the kernels that do not rely on lane-ordered LDS atomics "
                         "(what a device that fails the create-time check falls back to): same bytes, slower -- how much is what this measures")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="wall time of each multi-process CPU leg")
    ap.add_argument("--place-batches", type=int, default=12,
                    help="untimed batches run one at a time before the warm-up: the library's placement calibration (twelve large batches per context) happens here")
    ap.add_argument("--no-store-probe", action="store_true")
    ap.add_argument("--no-rank-parity", action="store_true", help="N > 1: skip the per-rank comparison of every rank's shard with the oracle (`parity_check_ranks` in the line)")
    ap.add_argument("--live-traffic", default="auto", choices=["auto", "on", "off"],
                    help="roofline.traffic measured by two rocprofv3 --pmc child passes of this script on this box (auto: with the CPU legs, "
                         "i.e. the default single-GPU run; the hash-matched profiles/traffic_latest.json is quoted otherwise)")
    ap.add_argument("--timing-every", type=int, default=3,
                    help="phase events (kernel_ms) on every n-th batch: each is a barrier packet between the kernels, 1.2 %% of a step "
                         "when every batch carries them (1: every batch; odd: the timed launches alternate between the context's two slots)")
    ap.add_argument("--pipeline-seconds", type=float, default=2.5,
                    help="wall time of the streaming leg (`pipeline` in the line: nothing staged ahead, sampler + staging + run + free "
                         "from one host thread); 0 skips it")
    ap.add_argument("--e2e-seconds", type=float, default=1.0,
                    help="wall time (at least two batches) of each end-to-end leg -- raw int16 into pinned host memory, svb-zd into pinned "
                         "host memory, BLOW5 into /dev/shm (`e2e` in the line; N = 1 only); 0 skips them")
    ap.add_argument("--e2e-batch-reads", type=int, default=2048, help="reads per batch of the end-to-end legs")
    ap.add_argument("--e2e-fast-batch-reads", type=int, default=8192, help="reads per batch of the stored-block BLOW5 leg (`e2e.blow5_fast`)")
    ap.add_argument("--small-batch-seconds", type=float, default=1.0,
                    help="wall time of each small-batch streaming leg (`small_batch` in the line; N = 1, worker-sharded runs of the genome "
                         "workloads only): the reference's default batch size, `-t 1 -K 1000` and `-t 8 -K 1000` (src/sim.c:208-209), "
                         "nothing staged ahead; 0 skips them")
    ap.add_argument("--every-batch-launches", type=int, default=24,
                    help="batches of an extra leg behind the timed region in which EVERY batch carries the phase events: kernel_ms over >= 20 "
                         "launches (`kernel_ms_every_batch`; the events cost ~1 %% of a step, so the timed region carries them on every "
                         "--timing-every'th batch only); 0 skips it")
    ap.add_argument("--workers-per-gpu", type=int, default=None,
                    help="W > 0: the job has T = N*W virtual workers, W per GPU (sharded by worker, no data-path collective), and every "
                         "batch of N*K reads is split over them as the reference's static partition does (src/thread.c:80-99): "
                         "`-t N*W -K N*K`; W = 1 on one GPU is the reference's reproducible `-t 1`.  0: one worker per read (T = K)")
    ap.add_argument("--job-workers", type=int, default=0,
                    help="T of the whole job with RANGE sharding: every GPU owns all T workers and generates a range of each batch's "
                         "reads; the per-stream sample counts are all-gathered over RCCL once per batch (include/sqg.h)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL) for one rank per GPU; gloo only to exercise the N>1 control flow on a box with fewer GPUs than "
                         "ranks (ranks then share GPUs)")
    ap.add_argument("--lib", default=None,
                    help="measure THIS shared library instead of the in-tree build (A/B runs).  The SQG_LIB environment variable alone "
                         "is refused: a bench line must say which library it timed")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed even for ONE rank (rendezvous on 127.0.0.1): the model broadcast, the barriers, the "
                         "reductions and -- with --job-workers -- the per-batch all-gather of the stream counts then run through the "
                         "chosen backend (nccl = RCCL) exactly as they do at N > 1 (tests: RCCL on a one-GPU box)")
    ap.add_argument("--numa-pin", default="auto", choices=["auto", "on", "off"],
                    help="keep the rank's host threads on the CPUs of its GPU's NUMA node (auto: when there is more than one rank and "
                         "each has a GPU of its own)")
    ap.add_argument("--genome-from-rank0", action="store_true",
                    help="hg38-r10 / synth-r10 at N > 1: rank 0 alone makes (a real run: loads) the genome and broadcasts the bytes to the other "
                         "GPUs -- one RCCL broadcast over xGMI -- instead of every rank synthesising its own copy (same bytes either way)")
    ap.add_argument("--digest", type=int, default=0,
                    help="D > 0 (tests): fetch every timed batch's signal and report, per batch, the sums of the reads' xxh64 digests "
                         "over D equal parts of the job's batch (N ranks report N*D/N parts each)")
    args = ap.parse_args()
    if args.lib:
        os.environ["SQG_LIB"] = os.path.abspath(args.lib)
    elif os.environ.get("SQG_LIB"):
        raise SystemExit("SQG_LIB is set but --lib was not given: bench.py measures the in-tree build unless told otherwise")

    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and world_env is None:
        # launched the way a 1-GPU run is: start the N ranks ourselves (one process per GPU, rendezvous on 127.0.0.1)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        rc = subprocess.call(cmd)
        sys.exit(rc)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(world_env or "1")
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: one rank per GPU")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the signal path has no CPU fallback")
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and world > ndev:
        raise SystemExit(f"{world} ranks but {ndev} GPUs (use --backend gloo to share GPUs in a dry run)")
    local_rank %= ndev
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    pin = None
    if args.numa_pin == "on" or (args.numa_pin == "auto" and world > 1 and args.backend == "nccl"):
        pin = pin_to_gpu_numa_node(local_rank, local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))   # before any helper thread exists
    use_dist = world > 1 or args.force_dist
    if use_dist:
        kw = {}
        if world_env is None:                                      # a single rank started by hand (--force-dist)
            kw = dict(init_method=f"tcp://127.0.0.1:{free_port()}", rank=0, world_size=1)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, **kw)
        else:
            dist.init_process_group("gloo", **kw)

    wl_profile, wl_flags, wl_mode, wl_w, wl_k, wl_desc = WORKLOADS[args.workload]
    K = args.batch_reads or wl_k
    W = wl_w if args.workers_per_gpu is None else args.workers_per_gpu
    if args.profile is None:
        args.profile = wl_profile
    prof, flags = profiles.get_profile(args.profile)
    flags |= wl_flags
    if args.order_free:
        flags |= profiles.SQ_ORDER_FREE
    k = profiles.default_kmer_size(flags)
    n_k = 1 << (2 * k)
    amode = api.MODE_EXACT if args.mode == "exact" else api.MODE_CERTIFIED
    # pore model: rank 0 owns it; RCCL broadcast to the other GPUs (the only collective on the worker-sharded path)
    if rank == 0:
        mean, stdv = model.synthetic_model(k)
    else:
        mean, stdv = np.zeros(n_k, np.float32), np.zeros(n_k, np.float32)
    if use_dist:
        mean, stdv = shard.broadcast_model(mean, stdv, src=0)

    range_mode = args.job_workers > 0
    if range_mode:
        W = 0
    if W and K % W:
        raise SystemExit("--workers-per-gpu must divide --batch-reads")
    T = args.job_workers if range_mode else (W * world if W else K * world)
    w_lo, w_hi = (0, T) if range_mode else shard.worker_range(rank, world, T)
    gen = api.SignalGenerator(prof, flags, k, mean, stdv, seed=42, num_workers=T, device=local_rank, mode=amode,
                              worker_lo=w_lo, worker_hi=w_hi)
    if range_mode:
        gen.set_range_mode(True)
    # staging shares a batch's per-read libm draws with helper threads: this rank's share of the CPUs the job may use (the boxes of
    # the pool: 16 for up to 8 ranks), four at most -- eight ranks x four threads would be twice the quota
    aff, quota = cpu_allowance()
    share = aff if (pin is not None and pin.get("pinned")) else aff // world      # (after pinning the affinity mask is this rank's own slice)
    if quota:
        share = min(share, int(quota) // world)
    cpus_per_rank = max(1, share)
    stage_threads = max(1, min(4, cpus_per_rank))
    gen.set_stage_threads(stage_threads)
    # the phase events behind kernel_ms are barrier packets between the kernels (include/sqg.h, sqg_set_phase_timing): every
    # --timing-every'th batch carries them (short runs: every batch, so that the timed region holds timed launches)
    timing_every = max(1, args.timing_every) if args.steps >= 2 * max(1, args.timing_every) else 1
    gen.set_phase_timing(timing_every)
    r_lo, r_hi = shard.read_range(rank, world, K * world)          # range mode: my reads of the job's K*world-read batches

    host_contigs = None                                             # a host copy of (a small version of) the genome, for the CPU legs
    sm = api.SAMPLE_RNA if wl_mode == "rna" else api.SAMPLE_DNA
    if args.workload in ("hg38-r10", "synth-r10"):
        mb = args.genome_mb if args.genome_mb is not None else (None if args.workload == "hg38-r10" else 64.0)
        if args.genome_from_rank0 and use_dist:
            lens, _ = genome_layout(mb)
            seq = synthetic_genome_device(mb, dev)[0] if rank == 0 else torch.empty(sum(lens), dtype=torch.uint8, device=dev)
            if args.backend == "nccl":
                dist.broadcast(seq, src=0)                             # RCCL over xGMI: the packed genome, once
            else:
                h = seq.cpu()
                dist.broadcast(h, src=0)
                seq.copy_(h)
                del h
        else:
            seq, lens = synthetic_genome_device(mb, dev)
        torch.cuda.synchronize()
        gen.load_genome_device(seq.data_ptr(), lens, args.rlen, sm)
        genome_bases = int(sum(lens))
        genome_dev = (seq, lens) if ((args.small_batch_seconds > 0 or not args.no_cpu_baseline) and world == 1 and not args.digest) else None   # (the -t 8 leg's context loads it again)
        del seq
        torch.cuda.empty_cache()
    else:
        genome_dev = None
        host_contigs = load_contigs(SEQUINS) if args.workload == "sequin-rna004" else load_contigs(GENOME)
        gen.load_genome(host_contigs, args.rlen, sm)
        genome_bases = sum(len(c) for c in host_contigs)
    # worker of this GPU's read i: one each (T = K), or the static partition of the job's N*K-read batch over T = N*W workers
    workers = np.arange(w_lo, w_hi, dtype=np.int32) if not W else (w_lo + np.arange(K, dtype=np.int32) // (K // W)).astype(np.int32)

    # the reads ARE the reference's: gen_read (src/genread.c) with `--seed 42 -r <rlen> -t T -K N*K` on the resident genome,
    # sampled on the device at staging time (outside the timed region)
    nsteps = args.warmup + args.steps
    # staged ahead: everything, up to STAGE_AHEAD batches (0.2 GB of HBM each at the headline size); a longer run stages the rest
    # as it goes -- batch i + STAGE_AHEAD when batch i is done and freed, with STAGE_AHEAD - 1 batches still queued on the device
    STAGE_AHEAD = 160

    def stage_one():
        return gen.sample(K * world, None, lo=r_lo, hi=r_hi) if range_mode else gen.sample(K, workers)

    # the context's placement calibration (squigulator_amd/csrc/h_run.h, place_calibrate): over its first twelve large batches the library times the
    # scatter pass itself on three more allocations of each slot's event records and keeps the best -- host synchronisations that belong to a context's start-up, not to the
    # timed region: these batches run, one at a time, before the warm-up (untimed, freed; the job's next reads follow them)
    n_place = args.place_batches if (not args.digest and not range_mode) else 0
    for _ in range(n_place):
        pb = stage_one().run().wait()
        pb.free()
    batches = [stage_one() for _ in range(min(nsteps, args.warmup + STAGE_AHEAD))]
    # ... and the batch BEHIND the last timed one: staged, never run.  Every timed run then finds a successor whose first event pass it
    # takes along (k_part_hand_count), as the first timed batch's pass was taken along by the last warm-up batch: the timed region holds
    # exactly K of everything (without it the K-th counting pass would be missing from it)
    tail = [stage_one()] if len(batches) == nsteps else []

    def sync_all():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    n_rows = T * n_k
    keep = []
    gen_has_streams = not (flags & (profiles.SQ_IDEAL | profiles.SQ_IDEAL_AMP))

    def run(b):
        """default: one asynchronous launch sequence.  Range mode: counts -> all-gather in range order -> the rest"""
        if not range_mode or not gen_has_streams:
            return b.run()
        mine = shard.counts_tensor(b.run_begin(), n_rows, dev)
        if use_dist:
            before, after = shard.exchange_counts(mine)
        else:
            before = after = torch.zeros_like(mine)
        torch.cuda.synchronize()
        keep.append((before, after))                               # alive until the batch has run
        del keep[:-4]
        return b.run_end(before.data_ptr(), after.data_ptr())

    for b in batches[:args.warmup]:
        run(b).wait()
    sync_all()
    t0 = time.perf_counter()
    sig_ms, ev_ms, lean_ms, lean_bytes = [], [], [], []
    samples = bases = reads = fallback = fallback_of = 0
    digests = []
    timed = batches[args.warmup:]
    for b in timed:
        run(b)                        # asynchronous: the steps are queued back to back (range mode: one exchange per step)
    to_stage = args.steps - len(timed)
    i = 0
    while i < len(timed):
        b = timed[i]
        b.wait()
        tm = gen.timing()
        if tm["total_ms"] > 0:         # this batch carried the phase events
            sig_ms.append(tm["samples_ms"]); ev_ms.append(tm["events_ms"])
            lean_ms.append(tm["lean_ms"] if tm["lean_ms"] > 0 else tm["samples_ms"])
            lean_bytes.append(2 * b.n_samples + b.n_bases + 24 * b.n_reads)
        samples += b.n_samples; bases += b.n_bases; reads += b.n_reads
        if tm["fallback_samples"] >= 0:           # (-1: the batch's counters were reused by a later batch before it was waited for)
            fallback += tm["fallback_samples"]; fallback_of += b.n_samples
        if to_stage > 0:              # a long run: this batch makes room for one more (the device has STAGE_AHEAD - 1 queued meanwhile)
            b.free()
            timed[i] = None
            nb = stage_one()
            if to_stage == 1:
                tail.append(stage_one())  # (the successor of the run's last batch, as above)
            run(nb)
            timed.append(nb)
            to_stage -= 1
        i += 1
    sync_all()
    dt = time.perf_counter() - t0

    if args.digest:
        # tests: per batch, the sum of the reads' xxh64 digests over each of this rank's D parts (parts of the JOB's batch)
        import xxhash
        for b in timed:
            # (the slabs of older batches have been reused: generate again, deterministically the same?  No: the context's
            # streams have moved on.  Digests therefore need steps <= 2, which the slabs still hold.)
            sig = b.signal()
            per = b.n_reads // args.digest
            parts = []
            for d in range(args.digest):
                acc = 0
                for i in range(d * per, (d + 1) * per):
                    acc = (acc + xxhash.xxh64(sig[b.sig_off[i]:b.sig_off[i + 1]].tobytes()).intdigest()) & 0xFFFFFFFFFFFFFFFF
                parts.append(acc)
            digests.append(parts)

    on_gpu = not use_dist or args.backend == "nccl"
    tot = torch.tensor([float(samples), float(reads), dt], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
    dt_min = dt
    if use_dist:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        mn = tot.clone()
        dist.all_reduce(mn, op=dist.ReduceOp.MIN)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dt_max, dt_min = float(mx[2]), float(mn[2])
        if args.digest:
            allq = [None] * world
            dist.all_gather_object(allq, digests)
            # rank order = order of the parts inside a batch
            digests = [sum((allq[r][s] for r in range(world)), []) for s in range(len(digests))]
    else:
        dt_max = dt
    tot_samples, tot_reads = float(tot[0]), float(tot[1])

    def free_all():
        nonlocal batches, timed, tail
        for b in batches[:args.warmup] + [b for b in timed if b is not None] + tail:
            b.free()
        batches, timed, tail = [], [], []

    # every batch with the phase events: kernel_ms over >= 20 launches (the timed region above carries them on every n-th batch only)
    every = None
    if args.every_batch_launches > 0 and not args.digest and not range_mode:
        free_all()
        gen.set_phase_timing(1)
        nb_e = args.every_batch_launches
        sync_all()
        eb = [stage_one() for _ in range(nb_e + 1)]                 # (+ the successor of the last one: every run carries a first pass)
        l_ms, e_ms = [], []
        te0 = time.perf_counter()
        for b in eb[:nb_e]:
            run(b)
        for b in eb[:nb_e]:
            b.wait()
            tm = gen.timing()
            if tm["total_ms"] > 0:
                l_ms.append(tm["lean_ms"] if tm["lean_ms"] > 0 else tm["samples_ms"]); e_ms.append(tm["events_ms"])
        sync_all()
        te = time.perf_counter() - te0
        for b in eb:
            b.free()
        gen.set_phase_timing(timing_every)
        if l_ms:
            every = {"launches": len(l_ms), "k_samples_lean": float(np.mean(l_ms)), "k_samples_lean_min": float(np.min(l_ms)),
                     "k_samples_lean_max": float(np.max(l_ms)), "event side (k_events, k_part_*)": float(np.mean(e_ms)),
                     "ms_per_step": te / nb_e * 1e3,
                     "what": "a leg of its own behind the timed region: every batch carries the phase events (barrier packets: ~1 % of a step)"}

    # the reference's default batch size, streaming (src/sim.c:208-209: -t 8 -K 1000): nothing staged ahead.  The -t 1 leg runs on the bench's
    # context; the -t 8 leg needs a context of its own and runs LAST, when the bench's context is closed: two live contexts of one process on
    # one GPU share the hardware queues (seven streams on four queues), and the second one's staging stream then waits behind the first one's
    # idle queues' turn -- its 1000-read batches staged in 0.51 instead of 0.25 ms (tools/two_ctx_probe.py)
    small = None
    small_on = args.small_batch_seconds > 0 and world == 1 and not range_mode and not args.digest and W == 1

    def small_leg(g2, t_small):
        g2.set_phase_timing(0)
        g2.set_stage_threads(0)                                # (automatic: a 1000-read batch's draws are not worth waking helpers for)
        wk = (np.arange(1000, dtype=np.int32) // (1000 // t_small)).clip(0, t_small - 1).astype(np.int32)
        st1 = lambda: g2.sample(1000, wk)
        warm = st1().run(); warm.wait(); warm.free()
        sync_all()
        r = pipeline_leg(st1, lambda b: b.run(), 64, seconds=args.small_batch_seconds)      # (on the clock: >= 64 batches, then until the time is up)
        sync_all()
        return {"value": r[0] / r[2], "unit": "samples/s", "reads_per_s": r[1] / r[2], "seconds": r[2], "batches": r[4],
                "ms_per_batch": r[2] / r[4] * 1e3, "host_stage_ms_per_batch": r[3] * 1e3}

    if small_on:
        free_all()
        small = {"reads_per_batch": 1000, "what": "streaming (one host thread: sample + stage batch i+2, queue batch i+1, wait for and free batch i), "
                 "the reference's default -K 1000 (src/sim.c:208-209); -t 8: the static partition of a batch over eight virtual workers, a context of its own"}
        if T == 1:
            small["-t 1 -K 1000"] = small_leg(gen, 1)
            gen.set_phase_timing(timing_every)
            gen.set_stage_threads(stage_threads)

    # the streaming leg: the same job with nothing staged ahead (the sampler's and the staging kernels share the GPU with the generator)
    pipe = None
    if args.pipeline_seconds > 0 and not args.digest:
        free_all()
        n_pipe = int(min(max(args.pipeline_seconds / max(dt_max / max(args.steps, 1), 1e-5), 8), 20000))
        sync_all()
        pipe = pipeline_leg(stage_one, run, n_pipe)
        sync_all()

    e2e = None
    if args.e2e_seconds > 0 and world == 1 and not range_mode and not args.digest:
        free_all()
        Ke = min(args.e2e_batch_reads, K)
        we = workers[:Ke] if not W else np.minimum(w_lo + np.arange(Ke, dtype=np.int32) // max(Ke // W, 1), w_hi - 1).astype(np.int32)
        sync_all()
        e2e = e2e_legs(gen, prof, flags, lambda: gen.sample(Ke, we), Ke, args.e2e_seconds, kinds=("pinned_int16", "pinned_svb", "blow5"))
        sync_all()
        # the stored-block writer at a batch size of its own (the zlib leg above needs seconds per batch: hence its small ones): the file
        # system is what bounds it (tools/io_probe.cpp: 6.9 GB/s into ONE file of a tmpfs on the pool's boxes), so per-batch overheads count
        Kf = min(args.e2e_fast_batch_reads, K)
        wf = workers[:Kf] if not W else np.minimum(w_lo + np.arange(Kf, dtype=np.int32) // max(Kf // W, 1), w_hi - 1).astype(np.int32)
        e2e.update({k2: v for k2, v in e2e_legs(gen, prof, flags, lambda: gen.sample(Kf, wf), Kf, args.e2e_seconds, kinds=("blow5_fast", "blow5_fast_4files")).items() if k2.startswith("blow5_fast")})
        sync_all()

    genome_host = None
    if genome_dev is not None and rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload == "hg38-r10":
        genome_host = (genome_dev[0].cpu().numpy(), list(genome_dev[1]))     # (for cpu_baseline.t1_full_genome: the reference on the very genome)
    # which stream hand-out the timed steps used: the one that relies on lane-ordered LDS atomics (probed at create time, re-checked on a
    # sample of every slice of every batch) or -- a device that fails the probe, or --order-free -- the claim protocol (1.3 ms per step slower)
    lds_order = None
    if rank == 0:
        bad_, used_ = gen.probe_lds_order(256, 4)
        lds_order = {"in_use": used_, "probe_mismatches": bad_}
    store_peak_GBps = None
    if rank == 0 and not args.no_store_probe:
        store_peak_GBps = gen.probe_store_bandwidth(1 << 30, 10) / 1e9
    if small_on and genome_dev is not None:
        free_all()
        gen.close()
        gen = None
        g8 = api.SignalGenerator(prof, flags, k, mean, stdv, seed=42, num_workers=8, device=local_rank, mode=amode)
        g8.load_genome_device(genome_dev[0].data_ptr(), genome_dev[1], args.rlen, sm)
        small["-t 8 -K 1000"] = small_leg(g8, 8)
        g8.close()
    genome_dev = None
    torch.cuda.empty_cache()

    ptot = torch.tensor(list(pipe[:3]) if pipe else [0.0, 0.0, 1.0], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
    hst = torch.tensor([pipe[3] if pipe else 0.0, -(pipe[3] if pipe else 0.0)], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
    if use_dist:
        pmx = ptot.clone()
        dist.all_reduce(pmx, op=dist.ReduceOp.MAX)
        dist.all_reduce(ptot, op=dist.ReduceOp.SUM)
        ptot[2] = pmx[2]
        dist.all_reduce(hst, op=dist.ReduceOp.MAX)                  # the slowest and (negated) the fastest rank's host time per batch
    host_stage_max, host_stage_min = float(hst[0]), -float(hst[1])

    # N > 1: every rank checks its own shard against the oracle (worker sharding; range sharding's exchange is tests/test_range_sharding.py's)
    rank_parity = None
    if use_dist and world > 1 and not range_mode and not args.no_rank_parity:
        pc_contigs = host_contigs if host_contigs is not None else synthetic_genome_host(8.0)
        per_worker = 4 if W else 1
        Tp = T if W else 16 * world                                # (T = K: sixteen one-read workers per rank instead of K)
        lo_p, hi_p = (w_lo, w_hi) if W else shard.worker_range(rank, world, Tp)
        mine = parity_check_rank(prof, flags, k, mean, stdv, pc_contigs, min(args.rlen, 4000), Tp, lo_p, hi_p, per_worker, amode, local_rank)
        t_me = torch.tensor(mine, dtype=torch.int64, device="cuda" if (on_gpu and args.backend == "nccl") else "cpu")
        t_all = [torch.zeros_like(t_me) for _ in range(world)]
        dist.all_gather(t_all, t_me)
        rank_parity = [[int(x) for x in t.cpu().tolist()] for t in t_all]

    if rank == 0:
        steps = max(args.steps, 1)
        alg_bytes = (2 * samples + bases + 24 * reads) / steps           # per k_samples_lean launch (this rank)
        k_ms = float(np.mean(lean_ms)) if lean_ms else None             # the dominant kernel alone
        if k_ms is not None and not k_ms > 0:
            k_ms = None                                                 # (no timed launch in the region: null in the line, never NaN)
        achieved = None if k_ms is None else (float(np.mean(lean_bytes)) if lean_bytes else alg_bytes) / (k_ms * 1e-3)    # the timed launches' own bytes over their mean duration
        ms_per_step = dt_max / steps * 1e3
        if range_mode:
            regime = (f"-t {T} -K {K * world} (range sharding: every GPU owns all {T} worker(s) and generates {K} reads of each "
                      f"batch; one all-gather of {4 * n_rows} B per batch)")
        elif W:
            regime = (f"-t {T} -K {K * world} ({W} virtual worker(s) per GPU, sharded by worker; the reference's static partition of "
                      f"each batch, {K} reads per GPU)")
        else:
            regime = f"-t {T} -K {T} (T=K virtual workers, {K} per GPU)"
        wkey = f"{args.workload}|{args.profile}|W={W}|batch_reads={K}|rlen={args.rlen}|mode={args.mode}"
        from squigulator_amd import build as _build
        lib_path = api.LOADED_PATH or _build.LIB
        lib_info = api.build_info(api.load_library(lib_path))      # what the loaded library says it was built from (stamped by build.py)
        out = {
            "metric": "simulated raw samples/sec",
            "value": tot_samples / dt_max,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64" if args.mode == "exact" else "f32+f64",
            "data": "synthetic",
            "config": {
                "placement_batches": n_place,
                "workload": f"{wl_desc}; -x {args.profile} --seed 42 -r {args.rlen}, {regime}, {args.steps} batches "
                            f"({int(tot_reads)} reads)",
                "genome_bases": genome_bases,
                "reads_per_step_per_gpu": K, "kmer_size": k, "mode": args.mode, "order_free": bool(args.order_free),
                "lds_ordered_hand_out": lds_order,
                "reads": "gen_read on the device-resident genome (library sampler), as the reference with these options",
                "pore_model": "synthetic stand-in table (built-in ONT tables absent from the reference mount)",
            },
            "library": {"path": os.path.relpath(lib_path, ROOT), "sha256_16": _build.file_hash(lib_path),
                        "source_hash": _build.source_hash(), "built_from": lib_info.get("source_hash"), "dev": lib_info.get("dev") == "1",
                        "in_tree": os.path.abspath(lib_path) == os.path.abspath(_build.LIB),
                        "stale": lib_info.get("source_hash") != _build.source_hash()},
            # the N ranks' own clocks around the same K steps (value uses the slowest) and what torch.distributed says the world is
            "ranks": {"world_size": dist.get_world_size() if use_dist else 1, "backend": dist.get_backend() if use_dist else None,
                      "numa_pin": pin, "genome": "broadcast from rank 0" if (args.genome_from_rank0 and use_dist) else "made on every rank",
                      "ms_per_step_min": dt_min / steps * 1e3, "ms_per_step_max": dt_max / steps * 1e3},
            "pipeline": None if pipe is None else {
                "value": float(ptot[0]) / float(ptot[2]), "unit": "samples/s", "reads_per_s": float(ptot[1]) / float(ptot[2]),
                "seconds": float(ptot[2]), "batches_per_gpu": n_pipe + 1, "ms_per_step": float(ptot[2]) / (n_pipe + 1) * 1e3,
                "host_stage_ms_per_batch": pipe[3] * 1e3, "host_stage_ms_per_batch_min": host_stage_min * 1e3,
                "host_stage_ms_per_batch_max": host_stage_max * 1e3, "stage_threads": stage_threads, "cpus_per_rank": cpus_per_rank,
                "vs_value": float(ptot[0]) / float(ptot[2]) / (tot_samples / dt_max),
                "what": "nothing staged ahead but one batch: one host thread per GPU samples (device-side gen_read) + stages batch i+2, queues batch i+1, "
                        "waits for batch i and frees it; two batches in flight, one staged, results left in HBM"},
            "e2e": e2e,
            "small_batch": small,
            "kernel_ms_every_batch": every,
            "reads_per_s": tot_reads / dt_max,
            # samples the fp32 path left to FP64, over the batches whose counters were still theirs when they were waited for
            "fp64_fixup_frac": (fallback / fallback_of if fallback_of else None) if args.mode == "certified" else None,
            "samples_per_step_per_gpu": samples / steps,
            # the step of rounds 1-4's lines was 16384 reads per GPU: this line's step time at that size, for comparing rounds (the headline workload runs 32768
            # reads per batch since the last session of round 5 -- `-t 1` does not know the batch size, tests/batchsize_hg38.py -- and 3.5-7 % faster per read)
            "ms_per_16384_reads": ms_per_step * 16384.0 / max(reads / steps, 1.0),
            "kernel_ms": {"k_samples_lean": k_ms,
                          # from the end of the event side to the batch's last kernel; the fix-ups run on their own stream next to
                          # the NEXT batch's event kernels and are stretched by them, so this span is longer than the kernels in it
                          "k_scan..k_fixup* (span; fix-ups overlap the next batch's event side)": float(np.mean(sig_ms)) if sig_ms else None,
                          "event side (k_events, k_part_*)": float(np.mean(ev_ms)) if ev_ms else None},
            "roofline": {"bound": "hbm", "achieved": None if achieved is None else achieved / 1e9, "peak": HBM_PEAK_BYTES_PER_S / 1e9,
                         "unit": "GB/s", "frac": None if achieved is None else achieved / HBM_PEAK_BYTES_PER_S,
                         "traffic": pmc_traffic(wkey),
                         "kernel_ms": k_ms,
                         "kernel_ms_launches": len(lean_ms),        # timed launches inside the timed region (every `timing_every`th batch)
                         "timing_every": timing_every,
                         "kernel": "k_samples_lean" if args.mode == "certified" else "k_samples<exact>",
                         "algorithmic_bytes_per_launch": alg_bytes,
                         # the same algorithmic bytes over the whole step (event side + sample kernels + gaps): what the job sees
                         "step_frac": alg_bytes / (ms_per_step * 1e-3) / HBM_PEAK_BYTES_PER_S,
                         "workload_key": wkey},
        }
        st = pmc_step_traffic(wkey)
        out["roofline"]["traffic_source"] = None if out["roofline"]["traffic"] is None else "profiles/traffic_latest.json (PMC passes of the same sources and workload)"
        want_live = args.live_traffic == "on" or (args.live_traffic == "auto" and not args.no_cpu_baseline and not args.digest)
        lt = live_traffic(args) if (want_live and world == 1) else None
        if lt is not None:
            kk = lt["kernels"].get("k_samples_lean") or lt["kernels"].get("k_samples")
            out["roofline"]["traffic_profile"] = out["roofline"]["traffic"]              # (the committed profile's figure, for comparison)
            out["roofline"]["traffic"] = kk["hbm_bytes_per_launch"]
            out["roofline"]["traffic_source"] = "live"
            out["roofline"]["traffic_live"] = {"FETCH_SIZE_KiB": kk["FETCH_SIZE_KiB"], "WRITE_SIZE_KiB": kk["WRITE_SIZE_KiB"], "launches": kk["launches"],
                                               "seconds": lt["seconds"], "write_size_unit_check": lt["write_size_unit_check"],
                                               "failed_passes": lt["failed_passes"], "what": lt["what"]}
            st = step_traffic_of(lt["kernels"])
        if st is not None:
            # all the HBM traffic of a step (PMC, every kernel of the timed region) over the step's time: how busy the memory is
            out["roofline"]["step_traffic"] = st
            out["roofline"]["step_traffic_frac"] = st / (ms_per_step * 1e-3) / HBM_PEAK_BYTES_PER_S
        if store_peak_GBps is not None:
            out["roofline"]["measured_store_peak_GBps"] = store_peak_GBps
        # what the kernel occupies besides HBM bytes (the profile's counters priced with this run's time); `bound` in `resources` names
        # the largest share, and roofline.bound says the same thing when resources were measured (the line says ONE
        # thing about its bound).  `achieved`/`peak`/`frac` are quoted against the HBM roofline whatever the bound:
        # `quoted_against` says so.
        out["roofline"]["resources"] = pmc_resources(wkey, k_ms, samples / steps, out["roofline"].get("measured_store_peak_GBps"))
        if lt is not None:
            kk = lt["kernels"].get("k_samples_lean") or lt["kernels"].get("k_samples")
            cal = lt["calib_rgather8_l2_req_per_cycle"]
            live_res = resources_from(kk, cal or stored_calibration(), k_ms, samples / steps, out["roofline"].get("measured_store_peak_GBps"),
                                      "live: rocprofv3 --pmc child passes of this run (roofline.traffic_live), priced with this run's kernel_ms; cal_rgather8 "
                                      + ("counted in the same run" if cal else "from profiles/traffic_latest.json (tools/bin/pmc_calib not built)"))
            if live_res is not None:
                out["roofline"]["resources_profile"] = out["roofline"]["resources"]
                out["roofline"]["resources"] = live_res
        out["roofline"]["quoted_against"] = "hbm"
        if out["roofline"]["resources"] is not None:
            out["roofline"]["bound"] = out["roofline"]["resources"]["bound"]
        if digests:
            out["digest"] = digests
        if rank_parity is not None:
            out["parity_check_ranks"] = {"equal": all(r[0] == 0 and r[1] > 0 for r in rank_parity), "reads_differing": [r[0] for r in rank_parity],
                                         "reads": [r[1] for r in rank_parity], "samples": [r[2] for r in rank_parity], "digest": ["%016x" % r[3] for r in rank_parity],
                                         "what": "every rank: its own workers' reads of a small job of this regime (two batches, carried state) against the oracle's run of "
                                                 "the whole job, every int16 (bench.parity_check_rank)"}
        if args.no_cpu_baseline or world > 1:
            out["cpu_baseline"] = None                       # the CPU legs: single-GPU run only
        else:
            # a host-side genome of the same layout (the CPU's per-sample cost does not depend on the genome's size)
            if host_contigs is None:
                host_contigs = synthetic_genome_host(48.0 if args.genome_mb is None else min(args.genome_mb, 48.0))
            one_worker = bool(W) and not range_mode
            n_chk = 600 if one_worker else min(4 * (os.cpu_count() or 8), 2000)
            out["parity_check"], port_rate = parity_check(prof, flags, k, mean, stdv, host_contigs, args.rlen, n_chk, one_worker, amode)
            ref = None
            if not (flags & profiles.SQ_RNA):                     # (the harness' timed mode draws DNA reads from a FASTA)
                with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as tmp:
                    fa = os.path.join(tmp, "g.fa")
                    with open(fa, "wb") as f:
                        for i, c in enumerate(host_contigs):
                            f.write(b">c%d\n" % i + c + b"\n")
                    fa_full = None
                    if genome_host is not None and shutil.disk_usage(tmp).free > 2 * genome_host[0].size:
                        fa_full = os.path.join(tmp, "full.fa")
                        with open(fa_full, "wb") as f:
                            at = 0
                            for i, n_c in enumerate(genome_host[1]):
                                f.write(b">c%d\n" % i); f.write(memoryview(genome_host[0][at:at + n_c])); f.write(b"\n")
                                at += n_c
                    ref = cpu_reference(prof, flags, k, mean, stdv, fa, args.rlen, seconds=args.cpu_seconds, full_fasta=fa_full)
            out["cpu_baseline"] = ref or {"value": port_rate, "unit": "samples/s", "cores": 1 if one_worker else min(os.cpu_count() or 1, 64),
                                          "kind": "port", "sample": f"oracle restatement on the {n_chk} reads of parity_check"}
        print(json.dumps(out))
    for b in batches[:args.warmup] + [b for b in timed if b is not None] + tail:
        b.free()
    if gen is not None:
        gen.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
