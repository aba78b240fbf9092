#!/bin/bash
# placement-aware allocation (h_common.h: alloc_placed): parity subset, then twelve alternating repetitions with and without (SQG_NO_PLACEMENT=1, development library)
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r5be; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_fuzz_parity.py tests/test_config2_hg38.py tests/test_full_size.py tests/test_two_contexts.py tests/test_abi.py -m gpu -q -x 2>&1 | tail -3 | tee $OUT/pytest.log
L=squigulator_amd/csrc/libsqg_hip_dev.so
REPS=12 bash tools/ab_env.sh "$L" "$L SQG_NO_PLACEMENT=1" 2>&1 | tee $OUT/ab.log
python - <<'PY'
import re, statistics
a, b = [], []
for ln in open('gpurun_out/r5be/ab.log'):
    m = re.search(r'lean ([\d.]+) ms  events ([\d.]+) ms  step ([\d.]+) ms', ln)
    if not m: continue
    (b if 'NO_PLACEMENT' in ln else a).append(tuple(float(x) for x in m.groups()))
for name, v in (('placed', a), ('as allocated', b)):
    print(name, 'n', len(v), 'lean median %.3f' % statistics.median(x[0] for x in v), 'events median %.3f (min %.3f max %.3f)' % (statistics.median(x[1] for x in v), min(x[1] for x in v), max(x[1] for x in v)),
          'step median %.3f mean %.3f (min %.3f max %.3f)' % (statistics.median(x[2] for x in v), statistics.mean(x[2] for x in v), min(x[2] for x in v), max(x[2] for x in v)))
PY
