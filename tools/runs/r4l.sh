mkdir -p gpurun_out/r4l
python tools/k1000_probe.py 1000 400; python tools/k1000_probe.py 1000 400
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 ) > gpurun_out/r4l/tests.log 2>&1
tail -4 gpurun_out/r4l/tests.log
( timeout 300 python bench.py --batch-reads 1000 --steps 200 --warmup 10 --no-cpu-baseline --no-store-probe --e2e-seconds 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); p=d['pipeline']
print('K1000 value %.4e ms/step %.4f pipeline %.4e vs %.3f host_stage_ms %.3f' % (d['value'], d['ms_per_step'], p['value'], p['vs_value'], p['host_stage_ms_per_batch']))" )
