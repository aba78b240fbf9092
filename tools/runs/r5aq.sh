#!/bin/bash
# round 5, last session: the distribution of the kernels' durations in the streaming 1000-read legs (k_part_scan<64, 4> averages 227 us in the default run's
# stats with a 7.8-ms maximum: where?)
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5aq; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-store-probe --pipeline-seconds 0 --e2e-seconds 0 --every-batch-launches 0 --live-traffic off > $OUT/bench.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, collections, statistics
rows = list(csv.DictReader(open('gpurun_out/r5aq/kt_kernel_trace.csv')))
t0 = min(int(r['Start_Timestamp']) for r in rows)
d = collections.defaultdict(list)
for r in rows:
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')
    d[k].append(((int(r['Start_Timestamp']) - t0) / 1e6, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
for k, v in sorted(d.items(), key=lambda kv: -sum(x[1] for x in kv[1])):
    du = sorted(x[1] for x in v)
    if len(du) < 50: continue
    q = lambda p: du[min(len(du) - 1, int(p * len(du)))]
    print(f"{k[:44]:44s} n {len(du):5d}  p10 {q(.1):8.1f}  p50 {q(.5):8.1f}  p90 {q(.9):8.1f}  p99 {q(.99):8.1f}  max {du[-1]:8.1f}  total ms {sum(du)/1e3:7.1f}")
# where the long k_part_scan launches are
for k in d:
    if k.startswith('k_part_scan<64, 4>'):
        longs = [(round(t, 1), round(x)) for t, x in d[k] if x > 500]
        print(k, 'launches > 500 us:', len(longs), longs[:30])
PY
rm -f $OUT/kt_kernel_trace.csv
