#!/bin/bash
# round 5, last session: a read that is longer than a link should be but fits one gets a link of its own, whole (h_stage.h): parity, then the step with and without
# (SQG_NO_WHOLE_LINKS=1, development library), four alternating repetitions
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r5aw; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_split_chains.py tests/test_fuzz_parity.py tests/test_hip_parity.py tests/test_config2_hg38.py tests/test_range_sharding.py tests/test_many_reads.py tests/test_two_contexts.py -m gpu -q -x 2>&1 | tail -3 | tee $OUT/pytest.log
L=squigulator_amd/csrc/libsqg_hip_dev.so
REPS=4 bash tools/ab_env.sh "$L" "$L SQG_NO_WHOLE_LINKS=1" 2>&1 | tee $OUT/ab.log
