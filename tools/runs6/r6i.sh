#!/bin/bash
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6i; mkdir -p $OUT
L=squigulator_amd/csrc/libsqg_hip_dev.so
for i in 1 2 3 4 5 6; do SQG_VERBOSE=1 timeout 300 python bench.py --lib $PWD/$L --steps 30 --no-cpu-baseline --no-store-probe --pipeline-seconds 0 --e2e-seconds 0 --small-batch-seconds 0 --every-batch-launches 0 2>&1 | grep -E "placement|^\{" | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('step %.3f ms  lean %.3f  events %.3f' % (d['ms_per_step'], d['kernel_ms']['k_samples_lean'], d['kernel_ms']['event side (k_events, k_part_*)']))
    else: print(ln.rstrip()[16:])
"; done 2>&1 | tee $OUT/verbose.log
