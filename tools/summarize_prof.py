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
         "command: `python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-store-probe` under",
         "`rocprofv3 --kernel-trace --stats` (durations) and three separate `--pmc` passes (counters).", "",
         "## kernel durations (--kernel-trace --stats)", "", "| kernel | calls | avg us | % |", "|---|---|---|---|"]
for r in csv.DictReader(open(os.path.join(src, "trace_kernel_stats.csv"))):
    lines.append(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")
lines += ["", "## PMC counters, average per dispatch (millions)", ""]
for f in ("pmc1", "pmc2", "pmc3"):
    path = os.path.join(src, f + "_counter_collection.csv")
    if not os.path.exists(path):
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kn, d in acc.items():
        if not any(t in kn for t in ("k_samples", "k_events", "k_dwell")):
            continue
        lines.append(f"* `{kn}` ({f}): " + ", ".join(f"{c}={sum(v) / len(v) / 1e6:.1f}" for c, v in sorted(d.items())))
open(dst + "_summary.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
