#!/bin/bash
# can a short probe see which mode a freshly allocated buffer will run the scatter pass in?  tools/place_probe: 12 allocations, three access patterns each; twice
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r5bd; mkdir -p $OUT
for rep in 1 2; do timeout 300 tools/bin/place_probe 12; done 2>&1 | tee $OUT/probe.log
