#!/bin/bash
# round 5: the stored-block BLOW5 writer on the GPU (tests), and the e2e legs
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r5f; mkdir -p $OUT
timeout 900 python -X faulthandler -m pytest tests/test_blow5.py tests/test_abi.py tests/test_cpu_backend.py -m gpu -q -x 2>&1 | tail -5
for K in 2048 8192; do
timeout 600 python bench.py --no-cpu-baseline --pipeline-seconds 0 --small-batch-seconds 0 --every-batch-launches 0 --steps 6 --warmup 2 --e2e-seconds 2 --e2e-batch-reads $K > $OUT/bench$K.json 2> $OUT/bench$K.err
python -c "
import json; d = json.load(open('gpurun_out/r5f/bench$K.json')); print($K, {k: ('%.3e' % v['value'] if isinstance(v, dict) else v) for k, v in d['e2e'].items() if k != 'what'})"
done
