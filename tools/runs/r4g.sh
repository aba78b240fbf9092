mkdir -p gpurun_out/r4g
export BENCH_ARGS="--pipeline-seconds 0 --e2e-seconds 0 --steps 10"
for rep in 1 2; do
for spec in "SQG_NO_PRECOUNT=1" "SQG_PHC_ABL=2" "SQG_PHC_ABL=1" "SQG_PHC_GRID=4" "SQG_PHC_GRID=3"; do
  r=$(env $spec timeout 300 python bench.py --lib $PWD/squigulator_amd/csrc/libsqg_hip_dev.so --no-cpu-baseline --no-store-probe $BENCH_ARGS 2>/dev/null | python tools/ab_line.py)
  echo "$spec: $r"
done
done > gpurun_out/r4g/ab2.log 2>&1
cat gpurun_out/r4g/ab2.log
