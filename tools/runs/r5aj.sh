#!/bin/bash
# round 5, last session: the grouped sample loop again (round 4: four steps, 77 VGPRs, +3 %) -- now also TWO steps per group (54 VGPRs: seven
# wavefronts per SIMD stay): parity of the 2-step variant, then base / 2 rows / 2 compiler order / 4 rows alternately
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r5aj; mkdir -p $OUT
SQG_LIB=$PWD/tools/var_b_g2rows.so timeout 900 python -m pytest tests/test_hip_parity.py tests/test_fuzz_parity.py tests/test_config2_hg38.py -m gpu -q -x 2>&1 | tail -3 | tee $OUT/pytest.log
REPS=3 bash tools/ab_step.sh 2>&1 | tee $OUT/ab.log
