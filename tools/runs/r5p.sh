#!/bin/bash
# round 5, final sources: rocprofv3 stats + PMC passes (tools/prof_pmc.sh -> gpurun_out/r05/summary), then the driver's command once more without the profiler
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
bash tools/prof_pmc.sh r05 2>&1 | tail -5
cd $GRAFT_REPO_ROOT
ls gpurun_out/r05/summary/
cat gpurun_out/r05/summary/*.err | tail -5
cp gpurun_out/r05/summary/traffic_latest.json profiles/traffic_latest.json 2>/dev/null      # (so that the bench below quotes it)
timeout 900 python bench.py > gpurun_out/r05/summary/r05_bench.json 2> gpurun_out/r05/summary/r05_bench.err; tail -2 gpurun_out/r05/summary/r05_bench.err
python -c "
import json; d = json.load(open('gpurun_out/r05/summary/r05_bench.json')); print(d['ms_per_step'], d['value'], d['kernel_ms'], json.dumps(d['roofline'], indent=1)[:3000])"
