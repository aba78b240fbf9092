#!/bin/bash
# round 5, last session: 32 partitions of 8192 streams (-DPART_SUB_BITS=13) instead of 64 of 4096: an item's state gather touches half as many lines, the
# scatter pass' runs are twice as long; the hand-out's LDS doubles (two workgroups per CU).  Parity of the variant, then the step alternately
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r5al; mkdir -p $OUT
SQG_LIB=$PWD/tools/var_b_sub13.so timeout 900 python -m pytest tests/test_hip_parity.py tests/test_fuzz_parity.py tests/test_config2_hg38.py -m gpu -q -x 2>&1 | tail -3 | tee $OUT/pytest.log
REPS=3 bash tools/ab_env.sh "tools/var_a_base.so" "tools/var_b_sub13.so" "tools/var_b_sub13.so SQG_PHC_GRID=2" 2>&1 | tee $OUT/ab.log
