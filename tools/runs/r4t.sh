( timeout 900 python -m pytest tests/test_precount.py tests/test_split_chains.py tests/test_long_reads.py -m gpu -q -x 2>&1 | tail -4 )
export BENCH_ARGS="--pipeline-seconds 0 --e2e-seconds 0 --steps 12"
for rep in 1 2; do
for e in "SQG_PHC_SPLIT=0" "SQG_PHC_SPLIT=30" "SQG_PHC_SPLIT=45" "SQG_PHC_SPLIT=60" "SQG_NO_PRECOUNT=1"; do
r=$(env $e timeout 300 python bench.py --lib $PWD/squigulator_amd/csrc/libsqg_hip_dev.so --no-cpu-baseline --no-store-probe $BENCH_ARGS 2>/dev/null | python tools/ab_line.py); echo "$e: $r"; done
done
