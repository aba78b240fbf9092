#!/bin/bash
# placement calibration (place_calibrate): what it decides (SQG_VERBOSE), parity subset, then alternating repetitions with and without it
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6h; mkdir -p $OUT
L=squigulator_amd/csrc/libsqg_hip_dev.so
for i in 1 2 3; do SQG_VERBOSE=1 timeout 300 python bench.py --lib $PWD/$L --no-cpu-baseline --no-store-probe --pipeline-seconds 0 --e2e-seconds 0 --small-batch-seconds 0 --every-batch-launches 0 2>&1 | grep -E "placement|^\{" | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('step %.3f ms  lean %.3f  events %.3f' % (d['ms_per_step'], d['kernel_ms']['k_samples_lean'], d['kernel_ms']['event side (k_events, k_part_*)']))
    else: print(ln.rstrip())
"; done 2>&1 | tee $OUT/verbose.log
timeout 900 python -m pytest tests/test_00_configs.py tests/test_hip_parity.py tests/test_config2_hg38.py tests/test_precount.py -m gpu -q -x 2>&1 | tail -3 | tee $OUT/pytest.log
REPS=${REPS:-8} bash tools/ab_env.sh "$L SQG_NO_PLACE=1" "$L SQG_PLACE=on" 2>&1 | tee $OUT/ab.log
