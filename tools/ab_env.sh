#!/bin/bash
# A/B of (library, environment) pairs in one call: tools/ab_env.sh "<lib> VAR=val ..." "<lib> ..." ; three rounds
for rep in 1 2 3; do
for spec in "$@"; do
  words=($spec); lib=${words[0]}; envs=("${words[@]:1}")
  r=$(env "${envs[@]}" SQG_LIB=$PWD/$lib timeout 300 python bench.py --no-cpu-baseline --no-store-probe $BENCH_ARGS 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lean %.3f events %.3f step %.3f  %.3e' % (d['kernel_ms']['k_samples_lean'], d['kernel_ms']['event side (k_events, k_part_*)'], d['ms_per_step'], d['value']))" 2>&1 | tail -1)
  echo "$spec: $r"
done
done
