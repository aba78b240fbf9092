mkdir -p gpurun_out/r4b
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > gpurun_out/r4b/tests.log 2>&1
bash tools/prof_pmc.sh r04a > gpurun_out/r4b/prof.log 2>&1
tail -8 gpurun_out/r4b/tests.log; ls gpurun_out/r04a/summary
