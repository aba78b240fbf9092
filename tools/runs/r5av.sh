#!/bin/bash
# round 5, last session: the link target of staging at 32768 reads per batch (SQG_SPLIT_CHAINS, development library): default (8192, capped by the events per link),
# 16384, 32768, 65536
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r5av; mkdir -p $OUT
L=squigulator_amd/csrc/libsqg_hip_dev.so
REPS=3 bash tools/ab_env.sh "$L" "$L SQG_SPLIT_CHAINS=16384" "$L SQG_SPLIT_CHAINS=32768" "$L SQG_SPLIT_CHAINS=65536" 2>&1 | tee $OUT/ab.log
