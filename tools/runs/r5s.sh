#!/bin/bash
# round 5: the driver's command on the final tree (bench.py restructured: the -t 8 leg last, on a context of its own), the contract test, the other workloads
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r5s; mkdir -p $OUT
timeout 900 python bench.py > $OUT/r05_bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err
python -c "
import json; d = json.load(open('gpurun_out/r5s/r05_bench.json')); print(d['ms_per_step'], d['value'], d['kernel_ms']['k_samples_lean'], d['small_batch'], d['e2e']['blow5_fast']['value'], d['e2e']['blow5']['value'], d['roofline']['resources']['bound'], d['roofline']['measured_store_peak_GBps'])"
timeout 900 python -m pytest tests/test_bench_multi_gpu.py -m gpu -q -x -k "contract" 2>&1 | tail -3
for w in ncov-r9 sequin-rna004; do
  timeout 600 python bench.py --workload $w --no-cpu-baseline --e2e-seconds 0 --small-batch-seconds 0 > $OUT/r05_bench_$w.json 2>> $OUT/bench.err
  python -c "
import json; d = json.load(open('gpurun_out/r5s/r05_bench_$w.json')); print('$w', d['ms_per_step'], d['value'], d['kernel_ms'], d['roofline']['frac'], d['pipeline']['vs_value'])"
done
