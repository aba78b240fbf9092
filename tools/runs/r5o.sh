#!/bin/bash
# round 5: the draw-ahead thread for up to 16 workers (-t 8 -K 1000 streaming), parity of the few-worker tests
cd "$(dirname "$0")/../.."
for T in 1 8 16; do timeout 300 python tools/k1000_probe.py 1000 400 256 $T 2>&1 | tail -1; done
timeout 900 python -m pytest tests/test_split_chains.py tests/test_precount.py tests/test_fuzz_parity.py tests/test_sampler.py -m gpu -q -x 2>&1 | tail -3
