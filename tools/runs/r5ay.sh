#!/bin/bash
# whole-read links only where the 14-bit cap sets the link length: the small-batch legs (1000 reads per batch: the link target sets it) must not change;
# the full default line with and without (development library), then the link tests
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r5ay; mkdir -p $OUT
L=$PWD/squigulator_amd/csrc/libsqg_hip_dev.so
for rep in 1 2; do
  for e in "" "SQG_NO_WHOLE_LINKS=1"; do
    env $e timeout 600 python bench.py --lib $L --no-cpu-baseline --no-store-probe --pipeline-seconds 1 --e2e-seconds 0 --every-batch-launches 0 --live-traffic off 2>/dev/null > $OUT/line.json
    python - "$e" <<'PY'
import json, sys
for ln in open('gpurun_out/r5ay/line.json'):
    if ln.startswith('{"metric"'):
        d = json.loads(ln)
        sb = d['small_batch']
        print(sys.argv[1] or 'whole links', 'value %.4e  step %.3f  pipeline %.4e  small -t1 %.4e (%.3f ms)  -t8 %.4e (%.3f ms)' % (d['value'], d['ms_per_step'], d['pipeline']['value'],
              sb['-t 1 -K 1000']['value'], sb['-t 1 -K 1000']['ms_per_batch'], sb['-t 8 -K 1000']['value'], sb['-t 8 -K 1000']['ms_per_batch']))
PY
  done
done 2>&1 | tee $OUT/ab.log
timeout 1200 python -m pytest tests/test_split_chains.py tests/test_fuzz_parity.py tests/test_hip_parity.py tests/test_config2_hg38.py tests/test_range_sharding.py tests/test_many_reads.py tests/test_two_contexts.py -m gpu -q -x 2>&1 | tail -3 | tee $OUT/pytest.log
