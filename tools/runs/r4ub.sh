#!/bin/bash
# instruction-form / instruction-order microbenchmarks (tools/ubench{4,5,6}_gen.py) on the box: bash tools/runs/r4ub.sh [4] [5] [6]
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for n in ${@:-4 5 6}; do
    hipcc --offload-arch=gfx950 -O2 -o /tmp/ubench$n tools/ubench$n.hip 2>/dev/null && timeout 600 /tmp/ubench$n | tee gpurun_out/ubench$n.txt
done
