#!/bin/bash
# the last sources once more: the whole suite, and the few-worker randomised test on every fall-back path over 250 new seeds each
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r5bj; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $OUT/pytest.log
for v in $(python -c "import sys; sys.path.insert(0,'tests'); import test_fuzz_parity as F; print(' '.join(x for x in F.VARIANTS if x != 'default'))"); do
  timeout 600 python tools/fuzz_more.py 60000 250 $v 2>&1 | tail -1
done | tee $OUT/fuzz.log
