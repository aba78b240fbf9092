#!/bin/bash
# round 5, first call: (1) the suite's core parity tests on the tree as it stands; (2) CU-masked streams: how k_samples_lean and the
# event side scale with the CUs they get ("n,same": ONE stream on 32 - n CUs per XCD), and the event side of batch i+1 next to the
# sample kernel of batch i on disjoint CUs ("n": n CUs per XCD for the event side)
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r5a; mkdir -p $OUT
D=squigulator_amd/csrc/libsqg_hip_dev.so
export BENCH_ARGS="--pipeline-seconds 0 --e2e-seconds 0 --small-batch-seconds 0 --every-batch-launches 0"
export REPS=2
timeout 300 python bench.py --lib $PWD/$D --no-cpu-baseline --no-store-probe --steps 6 --warmup 2 $BENCH_ARGS > $OUT/first.log 2>&1
grep -q '^{"metric"' $OUT/first.log || { echo "the plain bench run failed:"; tail -20 $OUT/first.log; exit 1; }
SQG_VERBOSE=1 SQG_CU_SPLIT=8 timeout 300 python bench.py --lib $PWD/$D --no-cpu-baseline --no-store-probe --steps 6 --warmup 2 $BENCH_ARGS 2>&1 | grep -v '^{' | grep "sqg\]" | head -60 > $OUT/map.log
bash tools/ab_env.sh "$D" "$D SQG_OVERLAP=1" "$D SQG_CU_SPLIT=4,same" "$D SQG_CU_SPLIT=8,same" "$D SQG_CU_SPLIT=16,same" \
   "$D SQG_CU_SPLIT=6" "$D SQG_CU_SPLIT=8" "$D SQG_CU_SPLIT=10" "$D SQG_CU_SPLIT=12" "$D SQG_CU_SPLIT=14" 2>&1 | tee $OUT/ab.log
