mkdir -p gpurun_out/r4j
( timeout 300 python bench.py --batch-reads 1000 --steps 200 --warmup 10 --no-cpu-baseline --no-store-probe --e2e-seconds 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); p=d['pipeline']
print('K1000 value %.4e ms/step %.4f pipeline %.4e vs %.3f host_stage_ms %.3f' % (d['value'], d['ms_per_step'], p['value'], p['vs_value'], p['host_stage_ms_per_batch']))" ) > gpurun_out/r4j/k1000.log 2>&1
( SQG_STAGE_TIMING=1 timeout 300 python bench.py --batch-reads 1000 --steps 4 --warmup 2 --no-cpu-baseline --no-store-probe --e2e-seconds 0 --pipeline-seconds 0 2>&1 | grep "^\[stage\]" | tail -40 ) > gpurun_out/r4j/stage_timing.log 2>&1
cat gpurun_out/r4j/k1000.log; tail -14 gpurun_out/r4j/stage_timing.log
( timeout 1500 python tools/soak_oracle.py --samples 1.05e11 --out gpurun_out/r4j/soak_r10.md 2>&1 | tail -12 ) > gpurun_out/r4j/soak.log 2>&1
tail -4 gpurun_out/r4j/soak.log
