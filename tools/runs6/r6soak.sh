#!/bin/bash
# round 6, final sources: the soaks of rounds 3-5 once more -- HIP (exact + certified) against the oracle on the host's glibc, certified against exact
# on the device at bench size, randomised configurations on new seeds (default paths and every fall-back path)
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6soak; mkdir -p $OUT; rm -f $OUT/*.md
python -c "from squigulator_amd import build; print('source_hash', build.source_hash())" | tee $OUT/hash.txt
timeout 400 python tools/soak_oracle.py --profile dna-r10-prom --seconds 150 --out $OUT/soak_oracle.md 2>&1 | tail -2
timeout 200 python tools/soak_oracle.py --profile dna-r9-prom --seconds 60 --out $OUT/soak_oracle.md 2>&1 | tail -1
timeout 200 python tools/soak_oracle.py --profile rna004-prom --seconds 60 --out $OUT/soak_oracle.md 2>&1 | tail -1
timeout 300 python tools/stress.py --workload hg38-r10 --samples 3e11 --out $OUT/stress.md 2>&1 | tail -1
timeout 700 python tools/fuzz_more.py 60000 1500 2>&1 | tail -2 | tee $OUT/fuzz.log
for v in order-free per-link-rows no-precount wg-per-link; do timeout 300 python tools/fuzz_more.py 70000 250 $v 2>&1 | tail -1 | tee -a $OUT/fuzz.log; done
cat $OUT/*.md
