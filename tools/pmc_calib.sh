#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of tools/pmc_calib.hip's known-byte patterns -> gpurun_out/<name>/calib.txt
# usage (GPU box): tools/pmc_calib.sh calib
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-calib}
mkdir -p $OUT
BIN=$GRAFT_REPO_ROOT/tools/bin/pmc_calib
cd /tmp && export TMPDIR=/tmp
$BIN 1 > $OUT/rates.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT -o fetch -- $BIN 2 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT -o write -- $BIN 2 > $OUT/write.log 2>&1
python3 - $OUT <<'PY' | tee $OUT/calib.txt
import csv, collections, sys, os, glob
out = sys.argv[1]
known = 4 * (1 << 27)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for tag, name in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    for path in glob.glob(os.path.join(out, "**", tag + "_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == name:
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][name].append(float(r["Counter_Value"]))
print("kernel, FETCH_SIZE KiB -> bytes / known, WRITE_SIZE KiB -> bytes / known   (known = %d B per launch)" % known)
for k in sorted(acc):
    f = acc[k].get("FETCH_SIZE", [0.0]); w = acc[k].get("WRITE_SIZE", [0.0])
    fm, wm = sum(f) / len(f) * 1024, sum(w) / len(w) * 1024
    print("%-28s FETCH %.4e B = %.3f x known   WRITE %.4e B = %.3f x known" % (k, fm, fm / known, wm, wm / known))
PY
cat $OUT/rates.txt
