#!/bin/bash
# round 5: k_samples_lean's jump table without the workgroup barrier: parity, A/B, and the trace
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r5m; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_fuzz_parity.py tests/test_config2_hg38.py -m gpu -q -x 2>&1 | tail -3
REPS=3 bash tools/ab_step.sh 2>&1 | tee $OUT/ab.log
grep "lean trace" <(python bench.py --lib $PWD/tools/var_t2_trace.so --no-cpu-baseline --no-store-probe --pipeline-seconds 0 --e2e-seconds 0 --small-batch-seconds 0 --every-batch-launches 0 2>&1 >/dev/null) | tee $OUT/trace.log
