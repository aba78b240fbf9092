#!/bin/bash
# round 5 (c): k_scan inside k_part_mid, k_items inside k_part_hist -- parity, then A/B against SQG_NO_FOLD=1 in one call
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r5g; mkdir -p $OUT
timeout 1200 python -X faulthandler -m pytest tests/test_hip_parity.py tests/test_fuzz_parity.py tests/test_config2_hg38.py tests/test_precount.py tests/test_split_chains.py tests/test_long_reads.py tests/test_range_sharding.py -m gpu -q -x 2>&1 | tail -5
D=squigulator_amd/csrc/libsqg_hip_dev.so
REPS=3 bash tools/ab_env.sh "$D" "$D SQG_NO_FOLD=1" 2>&1 | tee $OUT/ab.log
