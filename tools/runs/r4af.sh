#!/bin/bash
# kernel timeline of one timed step (from the end of one k_samples_lean to the end of the next): durations and gaps
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r4af; rm -rf $OUT; mkdir -p $OUT
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o tl -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-store-probe --pipeline-seconds 0 --e2e-seconds 0 --timing-every 1000 > $OUT/tl.log 2>&1
cd $R
python - <<'PY'
import csv, glob
rows = []
for f in glob.glob("gpurun_out/r4af/**/tl_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r["Queue_Id"], r["Grid_Size_X"], r["Workgroup_Size_X"]))
rows.sort()
lean = [i for i, r in enumerate(rows) if "k_samples_lean" in r[2] and r[1] - r[0] > 1_500_000]
lo, hi = lean[-4], lean[-3]
t0 = rows[lo][1]; prev_end = t0
tot = {}
for s, e, n, q, g, w in rows[lo + 1: hi + 1]:
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:7.1f}  q{q} grid {g} wg {w}  {n}")
    prev_end = max(prev_end, e)
print("step (lean end to lean end): %.1f us" % ((rows[hi][1] - t0) / 1e3))
PY
