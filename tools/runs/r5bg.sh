#!/bin/bash
# experiment build (tools/var_x_place.so): best-of-six placement of the buffers of 64 MiB or more against plain hipMalloc, sixteen alternating repetitions of the timed region
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r5bg; mkdir -p $OUT
L=tools/var_x_place.so
REPS=16 bash tools/ab_env.sh "$L SQG_PLACE_MIN=67108864" "$L X=1" 2>&1 | tee $OUT/ab.log
python - <<'PY'
import re, statistics
a, b = [], []
for ln in open('gpurun_out/r5bg/ab.log'):
    m = re.search(r'lean ([\d.]+) ms  events ([\d.]+) ms  step ([\d.]+) ms', ln)
    if not m: continue
    (a if 'PLACE_MIN' in ln else b).append(tuple(float(x) for x in m.groups()))
for name, v in (('placed', a), ('as allocated', b)):
    print(name, 'n', len(v), 'lean median %.3f mean %.3f' % (statistics.median(x[0] for x in v), statistics.mean(x[0] for x in v)),
          'events median %.3f mean %.3f (min %.3f max %.3f)' % (statistics.median(x[1] for x in v), statistics.mean(x[1] for x in v), min(x[1] for x in v), max(x[1] for x in v)),
          'step median %.3f mean %.3f (min %.3f max %.3f)' % (statistics.median(x[2] for x in v), statistics.mean(x[2] for x in v), min(x[2] for x in v), max(x[2] for x in v)))
PY
