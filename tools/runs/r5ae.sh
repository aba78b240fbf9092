#!/bin/bash
# round 5: what the store FORM costs -- tools/store_probe2 (short / dword / dwordx4, aligned or 2 bytes off, buffer stores with a dropped lane),
# and k_samples_lean with its int16 pairs stored as ALIGNED dwords (timing-only -DSQG_ABL_STORE2A) against -DSQG_ABL_STORE2 and the kernel as it is
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r5ae; mkdir -p $OUT
timeout 120 tools/bin/store_probe2 2>&1 | tee $OUT/store_probe2.log
REPS=3 bash tools/ab_step.sh 2>&1 | tee $OUT/ab.log
