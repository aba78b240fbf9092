#!/bin/bash
# round 5, last session: the suite, smoke() and the driver's default bench command on the libraries as build() makes them in a fresh container
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r5ah; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $OUT/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee $OUT/smoke.log
( time timeout 600 python bench.py ) > $OUT/bench.json 2> $OUT/bench.err
tail -4 $OUT/bench.err
python tools/ab_line.py < $OUT/bench.json
