export BENCH_ARGS="--pipeline-seconds 0 --e2e-seconds 0 --steps 16"
for rep in 1 2; do
for e in "SQG_VERBOSE=0" "SQG_OVERLAP=1" "SQG_OVERLAP=1 SQG_LEAN_DYNLDS=2048" "SQG_OVERLAP=1 SQG_LEAN_DYNLDS=4096" "SQG_OVERLAP=1 SQG_LEAN_DYNLDS=9600" "SQG_LEAN_DYNLDS=4096"; do
r=$(env $e timeout 300 python bench.py --lib $PWD/squigulator_amd/csrc/libsqg_hip_dev.so --no-cpu-baseline --no-store-probe $BENCH_ARGS 2>/dev/null | python tools/ab_line.py); echo "$e: $r"; done
done
