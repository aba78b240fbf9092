#!/bin/bash
# round 5: the whole GPU suite on the tree with run_impl split into plan builders + executor; then the step (A/B against nothing: a sanity check of the rate)
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r5j; mkdir -p $OUT
timeout 1700 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | tee $OUT/pytest.log
D=squigulator_amd/csrc/libsqg_hip_dev.so
REPS=2 bash tools/ab_env.sh "$D" "$D SQG_NO_FOLD=1" 2>&1 | tee $OUT/ab.log
