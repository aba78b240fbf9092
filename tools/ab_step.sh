#!/bin/bash
# A/B of whole-step time: tools/var_*.so, default bench steps, several repetitions; env passes through
BENCH_ARGS=${BENCH_ARGS:---pipeline-seconds 0 --e2e-seconds 0 --small-batch-seconds 0 --every-batch-launches 0}   # (the timed region alone unless told otherwise)
for rep in $(seq 1 ${REPS:-3}); do
for f in tools/var_*.so; do
  r=$(timeout 300 python bench.py --lib $PWD/$f --no-cpu-baseline --no-store-probe $BENCH_ARGS "$@" 2>/dev/null | python tools/ab_line.py)
  echo "$f $r"
done
done
