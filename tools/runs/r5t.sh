#!/bin/bash
# round 5: the BLOW5 writer's sharded stored mode (tests), the e2e legs
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r5t; mkdir -p $OUT
timeout 900 python -X faulthandler -m pytest tests/test_blow5.py tests/test_abi.py tests/test_cpu_backend.py tests/test_dropin.py -m gpu -q -x 2>&1 | tail -5
timeout 600 python bench.py --no-cpu-baseline --pipeline-seconds 0 --small-batch-seconds 0 --every-batch-launches 0 --steps 6 --warmup 2 --e2e-seconds 2 > $OUT/bench.json 2> $OUT/bench.err
python -c "
import json; d = json.load(open('gpurun_out/r5t/bench.json')); print({k: ('%.3e %.1f GB/s' % (v['value'], v['GBps']) if isinstance(v, dict) else v) for k, v in d['e2e'].items() if k != 'what'})"
