#!/bin/bash
# grouped sample loop: parity first, then the A/B of the two loops in one call
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r4y
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_fuzz_parity.py tests/test_config2_hg38.py -m gpu -q -x 2>&1 | tail -5
bash tools/ab_step.sh 2>&1 | tee gpurun_out/r4y/ab.log
