#!/bin/bash
# whole-read links against cut reads, twelve alternating repetitions of the timed region (the boxes' event side wanders by +-3 % between identical runs)
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r5ax; mkdir -p $OUT
L=squigulator_amd/csrc/libsqg_hip_dev.so
REPS=12 bash tools/ab_env.sh "$L" "$L SQG_NO_WHOLE_LINKS=1" 2>&1 | tee $OUT/ab.log
python - <<'PY'
import re, statistics
a, b = [], []
for ln in open('gpurun_out/r5ax/ab.log'):
    m = re.search(r'events ([\d.]+) ms  step ([\d.]+) ms', ln)
    if not m: continue
    (b if 'NO_WHOLE' in ln else a).append((float(m.group(1)), float(m.group(2))))
for name, v in (('whole-read links', a), ('cut reads', b)):
    print(name, 'n', len(v), 'events median %.3f mean %.3f' % (statistics.median(x[0] for x in v), statistics.mean(x[0] for x in v)),
          'step median %.3f mean %.3f' % (statistics.median(x[1] for x in v), statistics.mean(x[1] for x in v)))
PY
