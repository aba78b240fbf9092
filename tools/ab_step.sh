#!/bin/bash
# A/B of whole-step time: tools/var_*.so, default bench steps, several repetitions; env passes through
for rep in 1 2 3; do
for f in tools/var_*.so; do
  r=$(SQG_LIB=$PWD/$f timeout 300 python bench.py --no-cpu-baseline --no-store-probe "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step %.4f ms  %.4e samples/s' % (d['ms_per_step'], d['value']))" 2>&1 | tail -1)
  echo "$f $r"
done
done
