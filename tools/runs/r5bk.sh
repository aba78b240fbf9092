#!/bin/bash
# experiment build (tools/var_x_realloc.so, not committed code): every 16 runs ONE named buffer of each slot is allocated anew -- whose re-allocation flips the
# scatter pass' mode?  Per window of 16 runs and slot: the median of the pass' launches, under rocprofv3 --kernel-trace, 130 timed steps
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5bk; mkdir -p $OUT
export TMPDIR=/tmp
for what in ${WHATS:-none part evrec state lbase items}; do
  rm -rf $OUT/kt; mkdir -p $OUT/kt
  ( cd /tmp && SQG_REALLOC_WHAT=$what timeout 600 rocprofv3 --kernel-trace -d $OUT/kt -o kt --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --lib $GRAFT_REPO_ROOT/tools/var_x_realloc.so --no-cpu-baseline --no-store-probe --pipeline-seconds 0 --e2e-seconds 0 --small-batch-seconds 0 --every-batch-launches 0 --live-traffic off --steps 130 --warmup 14 > $OUT/bench_$what.log 2>&1 )
  python - $what <<'PY'
import csv, glob, statistics, sys
f = glob.glob('gpurun_out/r5bk/kt/**/*kernel_trace.csv', recursive=True) + glob.glob('gpurun_out/r5bk/kt/*kernel_trace.csv')
d = {}
for r in sorted(csv.DictReader(open(f[0])), key=lambda r: int(r['Start_Timestamp'])):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')
    d.setdefault(k, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k in ('k_part_events<0, 1>', 'k_samples_lean<false, 4>'):
    v = d[k]                                    # launch i = run index i (one per batch)
    rows = []
    for w0 in range(0, len(v) - 15, 16):
        w = v[w0:w0 + 16]
        rows.append(f"{statistics.median(w[2::2]):.0f}/{statistics.median(w[3::2]):.0f}")
    print(f"{sys.argv[1]:6s} {k.split('<')[0]:14s}", '  '.join(rows))
PY
done 2>&1 | tee $OUT/windows.log
rm -rf $OUT/kt
grep -h "^\[sqg\] run" $OUT/bench_evrec.log $OUT/bench_part.log 2>/dev/null | head -60
