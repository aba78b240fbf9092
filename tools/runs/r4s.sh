( timeout 900 python -m pytest tests/test_precount.py tests/test_split_chains.py -m gpu -q -x 2>&1 | tail -4 )
for e in "SQG_NO_PRECOUNT=1" "SQG_VERBOSE=0" "SQG_NO_PRECOUNT=1" "SQG_VERBOSE=0"; do
r=$(env $e timeout 300 python bench.py --lib $PWD/squigulator_amd/csrc/libsqg_hip_dev.so --workload ncov-r9 --workers-per-gpu 1 --no-cpu-baseline --no-store-probe --e2e-seconds 0 --pipeline-seconds 0 2>/dev/null | python tools/ab_line.py); echo "$e: $r"; done
