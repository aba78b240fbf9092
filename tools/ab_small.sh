export BENCH_ARGS="--batch-reads 1000 --steps 300 --warmup 5 --pipeline-seconds 0"
L=squigulator_amd/csrc/libsqg_hip.so
bash tools/ab_env.sh "$L" "$L SQG_SPLIT_CHAINS=4096" "$L SQG_SPLIT_CHAINS=2048" "$L SQG_SPLIT_CHAINS=1024" "$L SQG_PART_SLICE=10240" "$L SQG_PART_SLICE=12288 SQG_SPLIT_CHAINS=2048" "$L SQG_PART_SLICE=4096" "$L SQG_PART_SLICE=16384 SQG_SPLIT_CHAINS=4096" 2>&1 | sort | awk '{print}' 
export BENCH_ARGS="--batch-reads 4096 --steps 100 --warmup 5 --pipeline-seconds 0"
bash tools/ab_env.sh "$L" "$L SQG_SPLIT_CHAINS=4096" "$L SQG_PART_SLICE=16384" 2>&1 | sort
