#!/bin/bash
# round 5: the timed region's length -- 20 steps behind 3 warm-up steps (the default so far) against 100 behind 10 and 100 behind 3, alternately
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r5ag; mkdir -p $OUT
Q="--no-cpu-baseline --no-store-probe --pipeline-seconds 0 --e2e-seconds 0 --small-batch-seconds 0 --every-batch-launches 0"
for rep in 1 2 3; do
  for sw in "20 3" "100 10" "100 3" "200 10"; do
    set -- $sw
    r=$(timeout 300 python bench.py $Q --steps $1 --warmup $2 2>/dev/null | python tools/ab_line.py)
    echo "steps $1 warmup $2: $r"
  done
done 2>&1 | tee $OUT/ab.log
