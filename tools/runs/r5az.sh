#!/bin/bash
# round 5, the last sources: profiles (stats + PMC passes), the whole suite, smoke, the default line, 1500 seeds of the randomised parity tests
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r5az; mkdir -p $OUT
bash tools/prof_pmc.sh r05 2>&1 | tail -2
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $OUT/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee $OUT/smoke.log
( time timeout 600 python bench.py ) > $OUT/bench.json 2> $OUT/bench.err
tail -4 $OUT/bench.err; python tools/ab_line.py < $OUT/bench.json
timeout 900 python tools/fuzz_more.py 50000 1500 2>&1 | tail -3 | tee $OUT/fuzz.log
