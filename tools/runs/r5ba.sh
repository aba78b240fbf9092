#!/bin/bash
# whole-read links against cut reads, kernel by kernel: rocprofv3 --kernel-trace medians of the timed launches (development library), two alternating repetitions
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5ba; mkdir -p $OUT
export TMPDIR=/tmp
L=$PWD/squigulator_amd/csrc/libsqg_hip_dev.so
for rep in 1 2; do
for e in "X=1" "SQG_NO_WHOLE_LINKS=1"; do
  rm -rf $OUT/kt; mkdir -p $OUT/kt
  ( cd /tmp && env $e timeout 600 rocprofv3 --kernel-trace -d $OUT/kt -o kt --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --lib $L --no-cpu-baseline --no-store-probe --pipeline-seconds 0 --e2e-seconds 0 --small-batch-seconds 0 --every-batch-launches 0 --live-traffic off --steps 40 > $OUT/bench.log 2>&1 )
  python - "$e" <<'PY'
import csv, glob, sys, statistics
f = glob.glob('gpurun_out/r5ba/kt/**/*kernel_trace.csv', recursive=True) + glob.glob('gpurun_out/r5ba/kt/*kernel_trace.csv')
d = {}
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')
    d.setdefault(k, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
out = []
tot = 0.0
for k in ('k_part_mid', 'k_part_events<0, 1>', 'k_part_hist', 'k_part_scan<64, 16>', 'k_part_scan<64, 4>', 'k_part_hand_count<1, 0>', 'k_samples_lean<false, 4>', 'k_fixup'):
    if k not in d: continue
    big = [x for x in d[k] if x > 0.5 * max(d[k])]
    m = statistics.median(big)
    if k != 'k_fixup': tot += m
    out.append(f"{k.split('<')[0]} {m:.0f}")
print(f"{sys.argv[1]:24s}", '  '.join(out), f"  sum {tot:.0f} us")
PY
done
done 2>&1 | tee $OUT/ab.log
rm -rf $OUT/kt
