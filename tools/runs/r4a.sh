mkdir -p gpurun_out/r4a
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/r4a/tests.log 2>&1
( timeout 400 python bench.py 2> gpurun_out/r4a/bench.err ) > gpurun_out/r4a/bench.json
( timeout 300 python tools/soak_oracle.py --seconds 60 --out gpurun_out/r4a/soak.md 2>&1 | tail -30 ) > gpurun_out/r4a/soak.log 2>&1
tail -5 gpurun_out/r4a/tests.log; tail -c 1500 gpurun_out/r4a/bench.json; tail -5 gpurun_out/r4a/soak.log
