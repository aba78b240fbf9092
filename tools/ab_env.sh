#!/bin/bash
# A/B of (library, environment) pairs in one call: tools/ab_env.sh "<lib> VAR=val ..." "<lib> ..." ; three rounds
BENCH_ARGS=${BENCH_ARGS:---pipeline-seconds 0 --e2e-seconds 0 --small-batch-seconds 0 --every-batch-launches 0}   # (the timed region alone unless told otherwise)
for rep in $(seq 1 ${REPS:-3}); do
for spec in "$@"; do
  words=($spec); lib=${words[0]}; envs=("${words[@]:1}")
  r=$(env "${envs[@]}" timeout 300 python bench.py --lib $PWD/$lib --no-cpu-baseline --no-store-probe $BENCH_ARGS 2>/dev/null | python tools/ab_line.py)
  echo "$spec: $r"
done
done
