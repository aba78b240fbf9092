#!/bin/bash
# round 5: the driver's command with the CPU reference also on the full-size genome
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r5u; mkdir -p $OUT
S=$SECONDS; timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench.py took $((SECONDS - S)) s"; tail -3 $OUT/bench.err
python -c "
import json; d = json.load(open('gpurun_out/r5u/bench.json')); c = d['cpu_baseline']; print(d['ms_per_step'], {k: c[k] for k in ('value','t1','to_blow5','t1_full_genome','full_genome_bases','cores')}, d['e2e']['blow5_fast_4files']['value'], d['small_batch']['-t 8 -K 1000']['value'])"
