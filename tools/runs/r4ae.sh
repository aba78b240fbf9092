#!/bin/bash
# the headline regime itself (-t 1: ONE worker chain) against the oracle on glibc, as long as a single oracle thread allows
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r4ae
cp profiles/r04_soak.md gpurun_out/r4ae/soak_before.md
timeout 900 python tools/soak_oracle.py --workers 1 --reads-per-worker 2048 --seconds 600 --out gpurun_out/r4ae/soak_t1.md 2>&1 | tail -5
cat gpurun_out/r4ae/soak_t1.md | tail -3
