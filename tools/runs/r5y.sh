#!/bin/bash
# round 5, FINAL sources: the whole GPU suite, the rocprofv3 stats + PMC passes, the driver's command without the profiler, the other workloads
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p gpurun_out/r5y
timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee gpurun_out/r5y/pytest.log
rm -rf gpurun_out/r05
bash tools/prof_pmc.sh r05 2>&1 | tail -3
cd $GRAFT_REPO_ROOT
cat gpurun_out/r05/summary/*.err | tail -5
cp gpurun_out/r05/summary/traffic_latest.json profiles/traffic_latest.json
timeout 900 python bench.py > gpurun_out/r05/summary/r05_bench.json 2> gpurun_out/r05/summary/r05_bench.err; tail -2 gpurun_out/r05/summary/r05_bench.err
for w in ncov-r9 sequin-rna004; do
  timeout 600 python bench.py --workload $w --no-cpu-baseline --e2e-seconds 0 --small-batch-seconds 0 > gpurun_out/r05/summary/r05_bench_$w.json 2>> gpurun_out/r05/summary/r05_bench.err
done
python - <<'PY'
import json
d = json.load(open('gpurun_out/r05/summary/r05_bench.json'))
print(d['ms_per_step'], d['value'], d['kernel_ms'], d['roofline']['frac'], d['roofline']['resources'], d['small_batch'], {k: v['value'] for k, v in d['e2e'].items() if isinstance(v, dict)}, d['cpu_baseline']['t1_full_genome'], d['library'])
PY
