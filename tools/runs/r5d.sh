#!/bin/bash
# round 5: the driver's command with the new legs (every-batch kernel_ms, small batches, resources), and the bench-line contract test
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r5d; mkdir -p $OUT
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -3 $OUT/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5d/bench.json"))
print(d["ms_per_step"], d["value"], d["kernel_ms"], d["kernel_ms_every_batch"], d["small_batch"], d["pipeline"]["ms_per_step"], d["pipeline"]["vs_value"], d["roofline"]["resources"])
PY
timeout 900 python -m pytest tests/test_bench_multi_gpu.py -m gpu -q -x -k "contract" 2>&1 | tail -5
