#!/bin/bash
# round 5, last session: is the memory side of k_samples_lean bound by the FORM of its stores?  Timing-only builds (results wrong): without the arithmetic the
# kernel is on its memory side (round 4: 2.34 ms); on top of that, the int16 pairs as dwords (STORE2: half of them 2 bytes off), as aligned dwords (STORE2A), no stores
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r5ak; mkdir -p $OUT
REPS=2 bash tools/ab_step.sh --steps 20 --warmup 3 2>&1 | tee $OUT/ab.log
