timeout 900 python -m pytest tests/test_hip_parity.py tests/test_split_chains.py tests/test_long_reads.py tests/test_fuzz_parity.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2 3; do
TL_LIB=$PWD/tools/var_before.so bash tools/timeline.sh --pipeline-seconds 0 --steps 8 --warmup 3 --timing-every 1000 | head -1
bash tools/timeline.sh --pipeline-seconds 0 --steps 8 --warmup 3 --timing-every 1000 | head -1
done
