#!/bin/bash
# placement calibration, final form (best of four allocations of evrec per slot, the scatter pass as its own probe): alternating repetitions with and
# without it (development library, SQG_NO_PLACE=1), then the whole GPU suite in the driver's form
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6j; mkdir -p $OUT
L=squigulator_amd/csrc/libsqg_hip_dev.so
REPS=${REPS:-10} bash tools/ab_env.sh "$L SQG_NO_PLACE=1" "$L SQG_PLACE=on" 2>&1 | tee $OUT/ab.log
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $OUT/pytest.log
