#!/bin/bash
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6d; mkdir -p $OUT
for s in 1 2 3; do timeout 300 python tools/runs6/dbg_pair.py $s; done 2>&1 | grep -v "^   " | tee $OUT/dbg.log
REPS=${REPS:-2} bash tools/ab_step.sh 2>&1 | tee $OUT/ab.log
