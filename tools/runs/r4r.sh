mkdir -p gpurun_out/r4r
cp profiles/r04_soak.md gpurun_out/r4r/soak.md
( timeout 600 python tools/soak_oracle.py --profile dna-r9-prom --samples 2.5e10 --out gpurun_out/r4r/soak.md 2>&1 | tail -3 )
( timeout 600 python tools/soak_oracle.py --profile rna004-prom --samples 2.5e10 --out gpurun_out/r4r/soak.md 2>&1 | tail -3 )
( timeout 600 python tools/soak_oracle.py --profile rna-r9-prom --samples 1e10 --out gpurun_out/r4r/soak.md 2>&1 | tail -3 )
( timeout 900 python tools/stress.py --workload hg38-r10 --samples 2e11 --out gpurun_out/r4r/soak_cert.md 2>&1 | tail -2 )
