#!/bin/bash
# round 5: the precount / ABI / release-build tests on the tree with the advisor's fixes; then the CU-masked streams with the mask's real
# meaning (bit i = CU slot i / 8 of XCC i % 8): "n,same" -- everything on 32 - n CUs per XCC; "n" -- the event side on n, the samples on 32 - n
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r5c; mkdir -p $OUT
timeout 900 python -X faulthandler -m pytest tests/test_precount.py tests/test_abi.py tests/test_release_build.py tests/test_split_chains.py tests/test_hip_parity.py -m gpu -q -x 2>&1 | tail -15 > $OUT/pytest.log
tail -5 $OUT/pytest.log
D=squigulator_amd/csrc/libsqg_hip_dev.so
export BENCH_ARGS="--pipeline-seconds 0 --e2e-seconds 0 --small-batch-seconds 0 --every-batch-launches 0"
export REPS=2
bash tools/ab_env.sh "$D" "$D SQG_CU_SPLIT=2,same" "$D SQG_CU_SPLIT=8,same" "$D SQG_CU_SPLIT=16,same" \
   "$D SQG_CU_SPLIT=8" "$D SQG_CU_SPLIT=10" "$D SQG_CU_SPLIT=12" 2>&1 | tee $OUT/ab.log
