#!/bin/bash
# round 5 (a), priced by timing-only builds: the hand-out writes 16 B per slot, the lean kernel makes ONE 16-B gather per event
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r5h; mkdir -p $OUT
REPS=3 bash tools/ab_step.sh 2>&1 | tee $OUT/ab.log
