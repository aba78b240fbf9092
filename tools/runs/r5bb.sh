#!/bin/bash
# does the event side's run-to-run spread come from where a process' buffers land?  Six processes of the same command under rocprofv3 --kernel-trace: medians of
# the timed launches per kernel and per buffer slot (even / odd batches run in the context's two buffer sets)
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5bb; mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2 3 4 5 6; do
  rm -rf $OUT/kt; mkdir -p $OUT/kt
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $OUT/kt -o kt --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-store-probe --pipeline-seconds 0 --e2e-seconds 0 --small-batch-seconds 0 --every-batch-launches 0 --live-traffic off --steps 40 > $OUT/bench.log 2>&1 )
  python - <<'PY'
import csv, glob, statistics
f = glob.glob('gpurun_out/r5bb/kt/**/*kernel_trace.csv', recursive=True) + glob.glob('gpurun_out/r5bb/kt/*kernel_trace.csv')
d = {}
for r in sorted(csv.DictReader(open(f[0])), key=lambda r: int(r['Start_Timestamp'])):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')
    d.setdefault(k, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
out = []
for k in ('k_part_events<0, 1>', 'k_part_hist', 'k_part_hand_count<1, 0>', 'k_samples_lean<false, 4>'):
    v = d[k][8:]                                   # the timed steps (past the warm-up and the first batches)
    a, b = statistics.median(v[0::2]), statistics.median(v[1::2])
    out.append(f"{k.split('<')[0]} {a:.0f}/{b:.0f}")
print('  '.join(out))
PY
done 2>&1 | tee $OUT/slots.log
rm -rf $OUT/kt
