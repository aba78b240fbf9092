#!/bin/bash
# round 5, last session: which of k_samples_lean's two second-level look-ups is the 0.33 ms of -DSQG_ABL_NODEP?  Timing-only builds (results wrong): the pore-table row
# (NODEP_MODEL) or the stream state (NODEP_STATE) read at consecutive addresses instead of where the event's record points
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r5am; mkdir -p $OUT
REPS=2 bash tools/ab_step.sh --steps 20 --warmup 3 2>&1 | tee $OUT/ab.log
