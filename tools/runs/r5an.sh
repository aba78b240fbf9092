#!/bin/bash
# round 5, last session: two consecutive events' pore-table rows from one 64-byte line (timing-only -DSQG_ABL_PAIRROW: the access pattern of a table indexed by the
# 8 bases two successive 9-mers share) against the committed look-up and against both look-ups contiguous (NODEP)
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r5an; mkdir -p $OUT
REPS=3 bash tools/ab_step.sh --steps 20 --warmup 3 2>&1 | tee $OUT/ab.log
