#!/bin/bash
# round 5: the whole GPU suite on the final sources (run_impl and stage_common split)
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r5n; mkdir -p $OUT
timeout 1700 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | tee $OUT/pytest.log
