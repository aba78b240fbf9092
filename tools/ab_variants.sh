#!/bin/bash
# A/B of kernel build variants: tools/var_*.so (built by hand with -D flags); prints kernel ms per variant
for f in tools/var_*.so; do
  r=$(timeout 300 python bench.py --lib $PWD/$f --steps 6 --warmup 2 --no-cpu-baseline --no-store-probe "$@" 2>/dev/null | python tools/ab_line.py)
  echo "$f $r"
done
