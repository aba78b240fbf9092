#!/bin/bash
# round 5, last session: does the GPU need seconds of load before it runs at its rate?  5 against 300 warm-up steps (2 s of the same work), alternately; and the
# order effect itself: the same command six times
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r5at; mkdir -p $OUT
Q="--no-cpu-baseline --no-store-probe --pipeline-seconds 0 --e2e-seconds 0 --small-batch-seconds 0 --every-batch-launches 0 --live-traffic off"
for rep in 1 2 3 4; do
  for W in 5 300; do
    r=$(timeout 600 python bench.py $Q --warmup $W 2>&1 | python tools/ab_line.py)
    echo "warmup $W: $r"
  done
done 2>&1 | tee $OUT/ab.log
