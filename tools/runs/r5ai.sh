#!/bin/bash
# round 5, last session: roofline.traffic measured by the run itself -- the contract test, then the driver's default command
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r5ai; mkdir -p $OUT
timeout 900 python -m pytest tests/test_bench_multi_gpu.py -m gpu -x -q -k "contract" 2>&1 | tail -15 | tee $OUT/pytest.log
( time timeout 900 python bench.py ) > $OUT/bench.json 2> $OUT/bench.err
tail -6 $OUT/bench.err
python tools/ab_line.py < $OUT/bench.json
python - <<'PY'
import json
for ln in open('gpurun_out/r5ai/bench.json'):
    if ln.startswith('{"metric"'):
        d = json.loads(ln)
        r = d['roofline']
        print({k: r.get(k) for k in ('traffic', 'traffic_source', 'traffic_profile', 'traffic_live', 'step_traffic', 'step_traffic_frac')})
PY
