#!/bin/bash
# per-kernel average durations of any command (rocprofv3 --kernel-trace --stats): tools/kstats.sh <python script and args>
OUT=$GRAFT_REPO_ROOT/gpurun_out/kstats
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ks -- python $GRAFT_REPO_ROOT/"$@" > $OUT/ks.log 2>&1
tail -4 $OUT/ks.log
python - <<PY
import csv, glob
for f in glob.glob("$OUT/**/ks_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:10.1f} us  total {float(r['TotalDurationNs'])/1e6:9.2f} ms")
PY
