#!/bin/bash
# slices of the bucketed hand-out: half / twice / four times as long as the default (164 M events / 4096, rounded to 1024: 40960)
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r5z; mkdir -p $OUT
D=squigulator_amd/csrc/libsqg_hip_dev.so
REPS=2 bash tools/ab_env.sh "$D" "$D SQG_PART_SLICE=20480" "$D SQG_PART_SLICE=81920" "$D SQG_PART_SLICE=163840" 2>&1 | tee $OUT/ab.log
