#!/bin/bash
# per-slot kernel durations (the two slots of a context alternate) next to the slots' buffer addresses (SQG_VERBOSE)
OUT=$GRAFT_REPO_ROOT/gpurun_out/slot_probe
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SQG_VERBOSE=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o sp -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-store-probe --pipeline-seconds 0 "$@" > $OUT/sp.log 2>&1
grep "\[sqg\] batch" $OUT/sp.log
python - <<PY
import csv, glob, collections
per = collections.defaultdict(list)
for f in glob.glob("$OUT/**/sp_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if n.startswith("k_"):
            per[n].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
for n, v in per.items():
    v.sort()
    d = [x[1] for x in v][4:24]
    if len(d) < 20: continue
    ev, od = d[0::2], d[1::2]
    print(f"{n[:40]:40s} even {sum(ev)/len(ev):8.1f}  odd {sum(od)/len(od):8.1f} us")
PY
