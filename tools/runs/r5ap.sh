#!/bin/bash
# round 5, last session: reads per batch -- 16384 (the default of the headline workload), 32768, 65536: what the per-batch fixed costs are worth
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r5ap; mkdir -p $OUT
Q="--no-cpu-baseline --no-store-probe --pipeline-seconds 0 --e2e-seconds 0 --small-batch-seconds 0 --every-batch-launches 0 --live-traffic off"
for rep in 1 2; do
  for K in 16384 32768 65536; do
    r=$(timeout 600 python bench.py $Q --batch-reads $K --steps $((100 * 16384 / K)) --warmup 4 2>&1 | python tools/ab_line.py)
    echo "K $K: $r"
  done
done 2>&1 | tee $OUT/ab.log
