#!/bin/bash
# round 5, last session: the hand-out's slice length at 32768 reads per batch (8064 slices of 40960 events by default): 81920, 163840 (development library)
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r5as; mkdir -p $OUT
L=squigulator_amd/csrc/libsqg_hip_dev.so
REPS=3 bash tools/ab_env.sh "$L" "$L SQG_PART_SLICE=81920" "$L SQG_PART_SLICE=163840" "$L SQG_PART_SLICE=20480" 2>&1 | tee $OUT/ab.log
