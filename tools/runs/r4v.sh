mkdir -p gpurun_out/r4v
( timeout 900 python tools/soak_oracle.py --samples 4e10 --out gpurun_out/r4v/soak.md 2>&1 | tail -3 )
( timeout 600 python tools/soak_oracle.py --profile dna-r9-prom --samples 1.5e10 --out gpurun_out/r4v/soak.md 2>&1 | tail -3 )
( timeout 600 python tools/soak_oracle.py --profile rna004-prom --samples 1.5e10 --out gpurun_out/r4v/soak.md 2>&1 | tail -3 )
