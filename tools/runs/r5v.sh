#!/bin/bash
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r5v; mkdir -p $OUT
REPS=3 bash tools/ab_step.sh 2>&1 | tee $OUT/ab.log
