#!/bin/bash
# the whole GPU suite under each fall-back / forced setting of the few-worker paths (tests that set the same variable themselves keep
# their own value): tools/suite_variants.sh    (about 80 s per variant on one MI355X)
cd "$(dirname "$0")/.."
skip_ord="--deselect tests/test_split_chains.py::test_lds_atomics_are_served_in_lane_order"
skip_cmp="--deselect tests/test_split_chains.py::test_split_and_unsplit_runs_agree_at_size"
run() { echo "== $1"; shift; env "$@" 2>&1 | tail -1; }
run "order-free kernels"            SQG_PART_CLAIMS=1      python -m pytest tests -q -m gpu -x $skip_ord
run "per-link rows (round 1)"       SQG_NO_PART=1          python -m pytest tests -q -m gpu -x
run "workgroup-per-link passes"     SQG_PART_WG_EVENTS=1   python -m pytest tests -q -m gpu -x
run "1024-event slices"             SQG_PART_SLICE=1024    python -m pytest tests -q -m gpu -x
run "one piece per segment"         SQG_SPLIT_CHAINS=100000 python -m pytest tests -q -m gpu -x $skip_cmp
run "three links"                   SQG_SPLIT_CHAINS=3     python -m pytest tests -q -m gpu -x $skip_cmp
