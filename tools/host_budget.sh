#!/bin/bash
# The host at the node's real CPU budget (16 CPUs for 8 ranks = 2 per rank): bench.py's streaming leg (nothing staged ahead: one host thread
# samples, stages, queues, waits and frees) pinned to 1 / 2 / 4 / 16 CPUs with taskset -- the staging helpers follow the affinity mask
# (bench.py: cpus_per_rank -> sqg_set_stage_threads).  usage: bash tools/host_budget.sh <out dir>
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/host_budget}; mkdir -p $OUT
for n in 1 2 4 16; do
  last=$((n - 1))
  taskset -c 0-$last timeout 600 python bench.py --no-cpu-baseline --no-store-probe --e2e-seconds 0 --small-batch-seconds 0 --every-batch-launches 0 --pipeline-seconds 3 > $OUT/cpus$n.json 2> $OUT/cpus$n.err
done
python - $OUT <<'PY'
import json, sys
out = sys.argv[1]
print("| CPUs | staging threads | timed region (all staged ahead) ms / step | streaming ms / step | streaming / timed | host staging ms per batch |")
print("|---|---|---|---|---|---|")
for n in (1, 2, 4, 16):
    try:
        d = json.load(open(f"{out}/cpus{n}.json"))
    except Exception as e:
        print(f"| {n} | failed: {e} |"); continue
    p = d["pipeline"]
    print(f"| {n} | {p['stage_threads']} | {d['ms_per_step']:.3f} | {p['ms_per_step']:.3f} | {p['vs_value']:.3f} | {p['host_stage_ms_per_batch']:.3f} |")
PY
