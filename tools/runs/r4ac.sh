#!/bin/bash
# TA / TCP counters on tools/pmc_calib.hip's known access patterns: what is an "access", and how many does a CU take per cycle
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r4ac; rm -rf $OUT; mkdir -p $OUT
BIN=$PWD/tools/bin/pmc_calib
cd /tmp && export TMPDIR=/tmp
$BIN 1 > $OUT/rates.txt 2>&1
i=0
for set in "TCP_TOTAL_ACCESSES_sum GRBM_GUI_ACTIVE" "TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" "TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT -o p$i -- $BIN 2 > $OUT/p$i.log 2>&1
done
python3 - $OUT <<'PY' | tee $OUT/calib.txt
import csv, collections, sys, os, glob
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(out, "**", "p*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    d = {c: sum(v) / len(v) for c, v in acc[k].items()}
    cyc = d.get("GRBM_GUI_ACTIVE", 0) / 8
    ins = d.get("TA_FLAT_READ_WAVEFRONTS_sum", 0) + d.get("TA_FLAT_WRITE_WAVEFRONTS_sum", 0)
    print("%-26s cycles %.3e  accesses %.3e = %.2f / cycle / CU, %.1f per instruction (%.3e loads %.3e stores)  TA busy %.2f  TCP pending stall %.2f  L2 rd req %.3e wr req %.3e" % (
        k, cyc, d.get("TCP_TOTAL_ACCESSES_sum", 0), d.get("TCP_TOTAL_ACCESSES_sum", 0) / 256 / max(cyc, 1), d.get("TCP_TOTAL_ACCESSES_sum", 0) / max(ins, 1),
        d.get("TA_FLAT_READ_WAVEFRONTS_sum", 0), d.get("TA_FLAT_WRITE_WAVEFRONTS_sum", 0), d.get("TA_TA_BUSY_sum", 0) / 256 / max(cyc, 1),
        d.get("TCP_PENDING_STALL_CYCLES_sum", 0) / 256 / max(cyc, 1), d.get("TCP_TCC_READ_REQ_sum", 0), d.get("TCP_TCC_WRITE_REQ_sum", 0)))
PY
cat $OUT/rates.txt
