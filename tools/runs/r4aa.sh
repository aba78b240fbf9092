#!/bin/bash
# the end-to-end legs under the runtime's copy-engine settings: does the D2H copy have to be a blit kernel (it slows the next batch's kernels 100x while it runs)?
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r4aa
run() { echo "== $*"; env "$@" python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-store-probe --pipeline-seconds 0 --e2e-seconds 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])['e2e']
print({k: (round(v['value'] / 1e10, 3), round(v['bytes_per_sample'], 2)) for k, v in d.items() if isinstance(v, dict)})"; }
run A=1
run HSA_ENABLE_SDMA=1
run HSA_ENABLE_SDMA=0
run GPU_MAX_HW_QUEUES=8
