#!/bin/bash
# the randomised parity tests over new seeds under the fall-back settings of the few-worker paths (development library)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r4ak
for v in SQG_PART_CLAIMS=1 SQG_NO_PART=1 SQG_PART_WG_EVENTS=1 SQG_NO_PRECOUNT=1; do
  echo "== $v"; env $v timeout 400 python tools/fuzz_more.py 7000 500 2>&1 | tail -3
done 2>&1 | tee gpurun_out/r4ak/log.txt
