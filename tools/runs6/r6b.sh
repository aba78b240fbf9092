#!/bin/bash
# round 6: two consecutive samples per lane (-DSQG_LEAN_PAIR=1): parity of the variant (config tests + the parity files), then alternating repetitions of the timed region
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6b; mkdir -p $OUT
SQG_LIB=$PWD/tools/var_b_pair.so timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_fuzz_parity.py tests/test_config2_hg38.py tests/test_sampler.py -m gpu -q -x 2>&1 | tail -15 | tee $OUT/pytest.log
REPS=${REPS:-4} bash tools/ab_step.sh 2>&1 | tee $OUT/ab.log
