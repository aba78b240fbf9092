#!/bin/bash
# kernel timeline of one bench batch (rocprofv3 --kernel-trace): start offset, duration, gap to the previous kernel
OUT=$GRAFT_REPO_ROOT/gpurun_out/timeline
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -o tl -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-store-probe "$@" > $OUT/tl.log 2>&1
python - <<PY
import csv, glob
rows = []
for f in glob.glob("$OUT/**/tl_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:44]))
for f in glob.glob("$OUT/**/tl_memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
rows.sort()
# last full batch: from the last k_events on
idx = [i for i, r in enumerate(rows) if "k_events" in r[2]]
lo = idx[-2] if len(idx) > 1 else idx[-1]
hi = idx[-1]
prev_end = rows[lo][0]
t0 = rows[lo][0]
for s, e, n in rows[lo:hi + 1]:
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:7.1f}  {n}")
    prev_end = max(prev_end, e)
PY
