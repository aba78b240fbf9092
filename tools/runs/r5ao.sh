#!/bin/bash
# round 5, last session: the event-start table's pointer in a VGPR (-DSQG_LEAN_TBV=1: three VALU instructions less per four steps of k_samples_lean's loop);
# parity of the variant, then six alternating repetitions of the default timed region
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r5ao; mkdir -p $OUT
SQG_LIB=$PWD/tools/var_b_tbv.so timeout 900 python -m pytest tests/test_hip_parity.py tests/test_fuzz_parity.py tests/test_config2_hg38.py -m gpu -q -x 2>&1 | tail -3 | tee $OUT/pytest.log
REPS=6 bash tools/ab_step.sh 2>&1 | tee $OUT/ab.log
