mkdir -p gpurun_out/r4u
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 ) > gpurun_out/r4u/tests.log 2>&1
tail -3 gpurun_out/r4u/tests.log
bash tools/prof_pmc.sh r04 > gpurun_out/r4u/prof.log 2>&1
ls gpurun_out/r04/summary
( timeout 400 python bench.py 2> gpurun_out/r4u/bench.err ) > gpurun_out/r4u/bench.json
python tools/ab_line.py < gpurun_out/r4u/bench.json
