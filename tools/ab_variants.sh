#!/bin/bash
# A/B of kernel build variants: tools/var_*.so (built by hand with -D flags); prints k_signal ms per variant
for f in tools/var_*.so; do
  r=$(SQG_LIB=$PWD/$f timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-store-probe "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms  %.3e samples/s  frac %.4f' % (d['kernel_ms']['k_samples_lean'], d['value'], d['roofline']['frac']))")
  echo "$f $r"
done
