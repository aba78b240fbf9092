#!/bin/bash
# the round's profile on the final sources (after the k_link_prefix fix): stats + PMC passes, then the driver's bench command
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r4al
bash tools/prof_pmc.sh r04 > gpurun_out/r4al/prof.log 2>&1
ls gpurun_out/r04/summary
( timeout 300 python bench.py 2> gpurun_out/r4al/bench.err ) > gpurun_out/r4al/bench.json
python tools/ab_line.py < gpurun_out/r4al/bench.json
