#!/bin/bash
# scatter pass: part[] written in whole 128-B lines (rings of 64 slots: -DPEV_LINE=32) against 64-B lines (rings of 32)
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r5ac; mkdir -p $OUT
SQG_LIB=$PWD/tools/var_b_line32.so timeout 600 python -m pytest tests/test_fuzz_parity.py tests/test_split_chains.py tests/test_long_reads.py -m gpu -x -q 2>&1 | tail -3 | tee $OUT/pytest.log
REPS=3 bash tools/ab_step.sh 2>&1 | tee $OUT/ab.log
