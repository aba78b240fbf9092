#!/bin/bash
# one rocprofv3 --pmc pass of the bench: tools/pmc_quick.sh "<counters>" [bench args]
# (under a timeout: a counter set the hardware cannot collect aborts rocprofv3, which then hangs in its signal handler)
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcq
rm -rf $OUT; mkdir -p $OUT
C="$1"; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 ${PMC_TIMEOUT:-240} rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT -o q -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-store-probe "$@" > $OUT/q.log 2>&1
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/q_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if any(t in k for t in ("k_events", "k_samples_lean", "k_part")):
        print(k, {c: "%.4g" % (sum(v) / len(v)) for c, v in d.items()})
PY
