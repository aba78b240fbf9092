#!/bin/bash
# experiment build (tools/evpart_exp.patch): a slot's evrec and part[] in ONE allocation, SQG_EVPART_GAP bytes apart -- is the scatter pass' mode then the same in
# every process, and which gap gives the fast one?  Per-slot medians under rocprofv3 --kernel-trace, three processes per setting
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5bm; mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2 3; do
for gap in separate 0 65536 1114112 2162688; do
  rm -rf $OUT/kt; mkdir -p $OUT/kt
  ( cd /tmp && env $( [ $gap = separate ] && echo X=1 || echo SQG_EVPART_GAP=$gap ) timeout 600 rocprofv3 --kernel-trace -d $OUT/kt -o kt --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --lib $GRAFT_REPO_ROOT/tools/var_x_evpart.so --no-cpu-baseline --no-store-probe --pipeline-seconds 0 --e2e-seconds 0 --small-batch-seconds 0 --every-batch-launches 0 --live-traffic off --steps 40 > $OUT/bench_$gap.log 2>&1 )
  python - $gap <<'PY'
import csv, glob, statistics, sys
f = glob.glob('gpurun_out/r5bm/kt/**/*kernel_trace.csv', recursive=True) + glob.glob('gpurun_out/r5bm/kt/*kernel_trace.csv')
d = {}
for r in sorted(csv.DictReader(open(f[0])), key=lambda r: int(r['Start_Timestamp'])):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')
    d.setdefault(k, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
out = []
for k in ('k_part_events<0, 1>', 'k_part_hist', 'k_part_hand_count<1, 0>', 'k_samples_lean<false, 4>'):
    v = d[k][8:]
    a, b = statistics.median(v[0::2]), statistics.median(v[1::2])
    out.append(f"{k.split('<')[0]} {a:.0f}/{b:.0f}")
print(f"gap {sys.argv[1]:>9s}: ", '  '.join(out))
PY
done
done 2>&1 | tee $OUT/slots.log
rm -rf $OUT/kt
