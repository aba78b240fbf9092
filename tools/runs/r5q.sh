#!/bin/bash
# why -t 8 -K 1000 stages in 0.48 ms per batch inside bench.py and in 0.23 in tools/k1000_probe.py: genome size? a second context?
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r5q; mkdir -p $OUT
for T in 1 8; do
  timeout 300 python tools/k1000_probe.py 1000 400 3088 $T 2>&1 | tail -1
  SQG_STAGE_TIMING=1 timeout 300 python tools/k1000_probe.py 1000 60 3088 $T 2>&1 | grep "^\[stage\]" | tail -60 > $OUT/stage_t$T.log
  python - $OUT/stage_t$T.log <<'PY'
import sys, collections
acc = collections.defaultdict(list)
for ln in open(sys.argv[1]):
    p = ln.split()
    acc[" ".join(p[1:-2])].append(float(p[-2]))
print({k: round(sum(v) / len(v), 3) for k, v in acc.items()})
PY
done
