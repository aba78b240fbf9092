#!/bin/bash
# round 5: where an item of k_samples_lean spends its time (-DSQG_LEAN_TRACE: shader-clock stamps per wavefront and phase)
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r5k; mkdir -p $OUT
A="--no-cpu-baseline --no-store-probe --pipeline-seconds 0 --e2e-seconds 0 --small-batch-seconds 0 --every-batch-launches 0"
for v in a_base t2_trace a_base t2_trace; do
  timeout 300 python bench.py --lib $PWD/tools/var_$v.so $A > $OUT/$v.json 2> $OUT/$v.err
  echo "$v: $(python tools/ab_line.py < $OUT/$v.json)"
  grep "lean trace" $OUT/$v.err
done 2>&1 | tee $OUT/trace.log
