SQG_PART_CLAIMS=1 python -m pytest "tests/test_split_chains.py::test_order_free_kernels_are_selectable_through_the_cfg" -m gpu -q -x 2>&1 | grep -E "^E|assert|passed|failed" | head -8
SQG_PART_WG_EVENTS=1 python -m pytest "tests/test_split_chains.py::test_every_batch_samples_the_lane_order_and_fails_loudly" -m gpu -q -x 2>&1 | grep -E "^E|assert|passed|failed" | head -8
SQG_NO_PART=1 python -m pytest "tests/test_bench_multi_gpu.py::test_eight_ranks_equal_one" -m gpu -q -x 2>&1 | grep -E "^E|assert|passed|failed|Error" | head -12
