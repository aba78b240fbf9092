#!/usr/bin/env python3
"""Condense a tools/prof_pmc.sh output directory (rocprofv3 CSVs) into profiles/<name>_summary.md + the
kernel-stats CSV.  usage: python tools/summarize_prof.py gpurun_out/r01 profiles/r01"""
import collections
import csv
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
shutil.copy(os.path.join(src, "trace_kernel_stats.csv"), dst + "_kernel_stats.csv")
lines = ["# rocprofv3 summary (" + os.path.basename(src) + ")", "",
         "command: `python bench.py` (the driver's default run) under `rocprofv3 --kernel-trace --stats` (durations); the counters come from",
         "three separate `--pmc` passes of `python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-store-probe`.", "",
         "## kernel durations (--kernel-trace --stats)", "", "| kernel | calls | avg us | % |", "|---|---|---|---|"]
for r in csv.DictReader(open(os.path.join(src, "trace_kernel_stats.csv"))):
    lines.append(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")
# per-launch durations of the two big kernels: the stats average mixes warm-up launches, the timed steps and the smaller
# parity-check batch of the CPU leg; the bench line's roofline.kernel_ms is the mean over the timed steps only
tr = os.path.join(src, "trace_kernel_trace.csv")
if os.path.exists(tr):
    per = collections.defaultdict(list)
    kernel_names = set()
    for r in csv.DictReader(open(tr)):
        kernel_names.add(r["Kernel_Name"])
        for key in ("k_samples_lean", "k_events", "k_part_events", "k_part_hand", "k_part_hist"):
            if key in r["Kernel_Name"]:
                if key in ("k_events", "k_part_events"):   # the counting and the scatter pass are two instantiations
                    key = r["Kernel_Name"].split("(")[0].replace("void ", "")
                per[key].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    bench_line = None
    log = os.path.join(src, "trace.log")
    if os.path.exists(log):
        import json
        for ln in open(log):
            if ln.startswith('{"metric"'):
                bench_line = json.loads(ln)
    lines += ["", "## per-launch durations (us) in launch order", ""]
    for key, v in per.items():
        v.sort()
        d = [x[1] for x in v]
        n_place = int(bench_line["config"].get("placement_batches", 0)) if bench_line else 0     # (round 6: untimed batches in front of the warm-up, bench.py --place-batches)
        head = (n_place + bench_line["warmup"] + bench_line["steps"] + 8) if bench_line else 40   # the streaming leg adds hundreds of launches
        lines.append(f"* `{key}`: " + ", ".join(f"{x:.0f}" for x in d[:head])
                     + (f", … ({len(d) - head} more: the streaming leg and the CPU leg's parity batch; mean {sum(d[head:]) / len(d[head:]):.0f})" if len(d) > head else ""))
        if key.startswith("k_part_events<1, 0>") or key.startswith("k_part_events<2, 0>"):
            if any("k_part_hand_count" in r2 for r2 in kernel_names):
                lines.append("  * (with the next batch staged ahead -- the timed region -- this pass runs inside `k_part_hand_count`: the launches listed here are "
                             "each run's first batch, the end-to-end legs' 2048-read batches and the parity batch, not the timed steps)")
                continue
        if bench_line and len(d) >= n_place + bench_line["warmup"] + bench_line["steps"]:
            w, k = n_place + bench_line["warmup"], bench_line["steps"]
            timed = d[w:w + k]
            lines.append(f"  * launches {w + 1}..{w + k} are the timed steps: mean {sum(timed) / len(timed):.1f} us"
                         + (f"; the bench line of this very run reports kernel_ms = {1e3 * bench_line['roofline']['kernel_ms']:.1f} us (hipEvents)"
                            if key == "k_samples_lean" else ""))
    if bench_line:
        with open(dst + "_bench_under_rocprof.json", "w") as fh:
            json.dump(bench_line, fh)
        lines.append(f"* bench line of the profiled run: `{os.path.basename(dst)}_bench_under_rocprof.json` (value {bench_line['value']:.4g} {bench_line['unit']}, "
                     f"{bench_line['ms_per_step']:.3f} ms per step; the CPU legs are slowed by the profiler)")
lines += ["", "## PMC counters, average per dispatch (millions)", ""]
for f in ("pmc1", "pmc2", "pmc3"):
    path = os.path.join(src, f + "_counter_collection.csv")
    if not os.path.exists(path):
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kn, d in acc.items():
        if not any(t in kn for t in ("k_samples", "k_events", "k_dwell", "k_part")):
            continue
        lines.append(f"* `{kn}` ({f}): " + ", ".join(f"{c}={sum(v) / len(v) / 1e6:.1f}" for c, v in sorted(d.items())))
open(dst + "_summary.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
