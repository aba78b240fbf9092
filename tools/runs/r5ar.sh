#!/bin/bash
# round 5, last session: reads per batch, finer -- 32768 (the new default), 49152, 65536, 98304; three alternating repetitions, ~0.7 s timed each
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r5ar; mkdir -p $OUT
Q="--no-cpu-baseline --no-store-probe --pipeline-seconds 0 --e2e-seconds 0 --small-batch-seconds 0 --every-batch-launches 0 --live-traffic off"
for rep in 1 2 3; do
  for K in 32768 49152 65536 98304; do
    r=$(timeout 600 python bench.py $Q --batch-reads $K --steps $((100 * 32768 / K)) --warmup 4 2>&1 | python tools/ab_line.py)
    echo "K $K: $r"
  done
done 2>&1 | tee $OUT/ab.log
