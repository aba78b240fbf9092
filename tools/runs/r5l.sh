#!/bin/bash
# round 5: k_samples_lean with 2 / 4 / 8 items per wavefront (SQG_LEAN_GRID caps the grid; the workgroup prologue -- jump table to LDS, barrier -- is 12 % of a
# wavefront's life with one item each: profiles/r05_lean_trace.md)
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r5l; mkdir -p $OUT
D=squigulator_amd/csrc/libsqg_hip_dev.so
REPS=2 bash tools/ab_env.sh "$D" "$D SQG_LEAN_GRID=81000" "$D SQG_LEAN_GRID=40500" "$D SQG_LEAN_GRID=20250" "$D SQG_LEAN_GRID=10000" 2>&1 | tee $OUT/ab.log
