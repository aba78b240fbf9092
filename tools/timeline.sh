#!/bin/bash
# kernel timeline of one bench batch (rocprofv3 --kernel-trace): start offset, duration, gap to the previous kernel
OUT=$GRAFT_REPO_ROOT/gpurun_out/timeline
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -o tl -- python $GRAFT_REPO_ROOT/bench.py ${TL_LIB:+--lib $TL_LIB} --steps 3 --warmup 1 --no-cpu-baseline --no-store-probe "$@" > $OUT/tl.log 2>&1
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
# one full batch of the timed region: from the first event pass of the second-to-last timed batch to the next batch's first event pass
first = ("k_part_events<1", "k_part_events<2", "k_events")
idx = [i for i, r in enumerate(rows) if any(f in r[2] for f in first) and "k_part_events<0, 1>" not in r[2]]
lo = idx[-3] if len(idx) > 2 else idx[0]
hi = idx[-2] if len(idx) > 2 else idx[-1]
prev_end = rows[lo][0]
t0 = rows[lo][0]
for s, e, n in rows[lo:hi + 1]:
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:7.1f}  {n}")
    prev_end = max(prev_end, e)
PY
