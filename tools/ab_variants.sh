#!/bin/bash
# A/B of kernel build variants: tools/var_*.so (built by hand with -D flags); prints kernel ms per variant
for f in tools/var_*.so; do
  r=$(SQG_LIB=$PWD/$f timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-store-probe "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lean %.3f ms  events %.3f ms  step %.3f ms  %.3e samples/s' % (d['kernel_ms']['k_samples_lean'], d['kernel_ms']['event side (k_events, k_part_*)'], d['ms_per_step'], d['value']))" 2>&1 | tail -1)
  echo "$f $r"
done
