#!/bin/bash
# the round's closing run: the whole GPU suite, then the driver's bench command
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r4ad
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
python bench.py > gpurun_out/r4ad/bench.json 2> gpurun_out/r4ad/bench.err
python tools/ab_line.py < gpurun_out/r4ad/bench.json
python -c "
import json
d = json.loads(open('gpurun_out/r4ad/bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step', 'vs_baseline', 'dtype')})
print(d['roofline']); print(d.get('e2e')); print(d.get('pipeline')); print(d['cpu_baseline']['value'], d['library'])"
