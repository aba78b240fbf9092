#!/bin/bash
# round 5: what a CU mask's bits stand for (tools/cumask_probe.hip); the fall-back variants of the fuzz / two-context tests and the
# precount tests on the tree with the advisor's fixes
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r5b; mkdir -p $OUT
timeout 120 tools/bin/cumask_probe 2>&1 | tee $OUT/cumask.log
timeout 1400 python -m pytest tests/test_fuzz_parity.py tests/test_two_contexts.py tests/test_precount.py tests/test_abi.py tests/test_release_build.py -m gpu -q -x 2>&1 | tail -8 | tee $OUT/pytest.log
