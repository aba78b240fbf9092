#!/bin/bash
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r5w; mkdir -p $OUT
timeout 700 python tools/fuzz_more.py 20000 1500 2>&1 | tail -2 | tee $OUT/fuzz.log
for v in order-free per-link-rows no-precount wg-per-link; do timeout 300 python tools/fuzz_more.py 30000 250 $v 2>&1 | tail -1 | tee -a $OUT/fuzz.log; done
