mkdir -p gpurun_out/r4o
export BENCH_ARGS="--pipeline-seconds 0 --e2e-seconds 0 --steps 12"
for rep in 1 2; do
for spec in "squigulator_amd/csrc/libsqg_hip_dev.so SQG_VERBOSE=0" "tools/var_items4.so SQG_VERBOSE=0" "squigulator_amd/csrc/libsqg_hip_dev.so SQG_LEAN_GRID=1792" "squigulator_amd/csrc/libsqg_hip_dev.so SQG_LEAN_GRID=3584"; do
  words=($spec); lib=${words[0]}; envs=("${words[@]:1}")
  r=$(env "${envs[@]}" timeout 300 python bench.py --lib $PWD/$lib --no-cpu-baseline --no-store-probe $BENCH_ARGS 2>/dev/null | python tools/ab_line.py)
  echo "$spec: $r"
done
done > gpurun_out/r4o/ab.log 2>&1
cat gpurun_out/r4o/ab.log
