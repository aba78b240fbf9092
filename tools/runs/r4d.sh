mkdir -p gpurun_out/r4d
TL_LIB=$PWD/squigulator_amd/csrc/libsqg_hip_dev.so bash tools/timeline.sh --pipeline-seconds 0 --e2e-seconds 0 > gpurun_out/r4d/timeline_fused.txt 2>&1
cat gpurun_out/r4d/timeline_fused.txt | tail -25
export BENCH_ARGS="--pipeline-seconds 0 --e2e-seconds 0"
for rep in 1 2; do
for spec in "squigulator_amd/csrc/libsqg_hip_dev.so SQG_NO_PRECOUNT=1" "squigulator_amd/csrc/libsqg_hip_dev.so SQG_VERBOSE=0" "tools/var_prio.so SQG_VERBOSE=0" "tools/var_cw2prio.so SQG_VERBOSE=0" "tools/var_cw2.so SQG_VERBOSE=0"; do
  words=($spec); lib=${words[0]}; envs=("${words[@]:1}")
  r=$(env "${envs[@]}" timeout 300 python bench.py --lib $PWD/$lib --no-cpu-baseline --no-store-probe $BENCH_ARGS 2>/dev/null | python tools/ab_line.py)
  echo "$spec: $r"
done
done > gpurun_out/r4d/ab.log 2>&1
cat gpurun_out/r4d/ab.log
