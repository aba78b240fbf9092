#!/bin/bash
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6c; mkdir -p $OUT
for s in 1 2; do timeout 300 python tools/runs6/dbg_pair.py $s; done 2>&1 | tail -120 | tee $OUT/dbg.log
