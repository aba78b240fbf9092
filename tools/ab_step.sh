#!/bin/bash
# A/B of whole-step time: tools/var_*.so, default bench steps, several repetitions; env passes through
for rep in $(seq 1 ${REPS:-3}); do
for f in tools/var_*.so; do
  r=$(timeout 300 python bench.py --lib $PWD/$f --no-cpu-baseline --no-store-probe "$@" 2>/dev/null | python tools/ab_line.py)
  echo "$f $r"
done
done
