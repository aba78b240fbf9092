mkdir -p gpurun_out/r4c
( timeout 900 python -m pytest tests/test_precount.py -x -q 2>&1 | tail -30 ) > gpurun_out/r4c/precount.log 2>&1
tail -5 gpurun_out/r4c/precount.log
export BENCH_ARGS="--pipeline-seconds 1.0 --e2e-seconds 0"
( bash tools/ab_env.sh "squigulator_amd/csrc/libsqg_hip_dev.so SQG_NO_PRECOUNT=1" "squigulator_amd/csrc/libsqg_hip_dev.so SQG_VERBOSE=0" ) > gpurun_out/r4c/ab.log 2>&1
cat gpurun_out/r4c/ab.log
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -30 ) > gpurun_out/r4c/tests.log 2>&1
tail -5 gpurun_out/r4c/tests.log
