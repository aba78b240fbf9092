#!/bin/bash
# experiment build (tools/var_x_place.so): best-of-six placement ONLY for the buffers the re-allocation experiment (tools/runs/r5bk.sh) found to decide the scatter pass'
# mode -- evrec (3.27 GB) and part / state (1.63 GB): 1.5-3.5 GB; the probe over the whole buffer, or over its first 40 % (evrec32 uses the first half of evrec);
# sixteen alternating repetitions of the timed region
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r5bl; mkdir -p $OUT
L=tools/var_x_place.so
REPS=16 bash tools/ab_env.sh "$L SQG_PLACE_MIN=1500000000 SQG_PLACE_MAX=3500000000" "$L SQG_PLACE_MIN=1500000000 SQG_PLACE_MAX=3500000000 SQG_PLACE_FRAC=0.4" "$L X=1" 2>&1 | tee $OUT/ab.log
python - <<'PY'
import re, statistics
g = {}
for ln in open('gpurun_out/r5bl/ab.log'):
    m = re.search(r'lean ([\d.]+) ms  events ([\d.]+) ms  step ([\d.]+) ms', ln)
    if not m: continue
    key = 'frac 0.4' if 'FRAC' in ln else 'placed' if 'PLACE_MIN' in ln else 'as allocated'
    g.setdefault(key, []).append(tuple(float(x) for x in m.groups()))
for name, v in g.items():
    print(name, 'n', len(v), 'lean median %.3f' % statistics.median(x[0] for x in v),
          'events median %.3f mean %.3f (min %.3f max %.3f)' % (statistics.median(x[1] for x in v), statistics.mean(x[1] for x in v), min(x[1] for x in v), max(x[1] for x in v)),
          'step median %.3f mean %.3f (min %.3f max %.3f)' % (statistics.median(x[2] for x in v), statistics.mean(x[2] for x in v), min(x[2] for x in v), max(x[2] for x in v)))
PY
