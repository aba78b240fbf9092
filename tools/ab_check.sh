#!/bin/bash
# parity of every tools/var_*.so (the golden-vector and fuzz tests through SQG_LIB), the swept error bound of each, then tools/ab_step.sh
for f in tools/var_*.so; do
  echo "== $f"
  SQG_VERBOSE=1 SQG_LIB=$PWD/$f timeout 900 python -m pytest tests/test_hip_parity.py tests/test_fuzz_parity.py -m gpu -x -q 2>&1 | grep -E "certified fp32|passed|failed|Error" | sort | uniq -c | sort -rn | head -4
done
bash tools/ab_step.sh "$@"
