mkdir -p gpurun_out/r4i
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 ) > gpurun_out/r4i/tests.log 2>&1
tail -4 gpurun_out/r4i/tests.log
( timeout 400 python bench.py 2> gpurun_out/r4i/bench.err ) > gpurun_out/r4i/bench.json
python tools/ab_line.py < gpurun_out/r4i/bench.json
python -c "
import json; d=json.load(open('gpurun_out/r4i/bench.json')); print(d['pipeline']['value'], d['pipeline']['vs_value'], d['pipeline']['host_stage_ms_per_batch'], d['e2e']['blow5']['value'], d['parity_check'])"
