#!/bin/bash
# round 6, final sources with the placement calibration (688b05e142268775): longer soaks
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6soak2; mkdir -p $OUT; rm -f $OUT/*.md
python -c "from squigulator_amd import build; print('source_hash', build.source_hash())" | tee $OUT/hash.txt
timeout 900 python tools/soak_oracle.py --profile dna-r10-prom --seconds 500 --out $OUT/soak_oracle.md 2>&1 | tail -1
timeout 400 python tools/soak_oracle.py --profile rna004-prom --seconds 120 --out $OUT/soak_oracle.md 2>&1 | tail -1
timeout 600 python tools/stress.py --workload hg38-r10 --samples 1e12 --out $OUT/stress.md 2>&1 | tail -1
timeout 300 python tools/stress.py --workload ncov-r9 --samples 3e11 --out $OUT/stress.md 2>&1 | tail -1
timeout 300 python tools/stress.py --workload sequin-rna004 --samples 3e11 --out $OUT/stress.md 2>&1 | tail -1
timeout 900 python tools/fuzz_more.py 80000 3000 2>&1 | tail -1 | tee $OUT/fuzz.log
cat $OUT/*.md
