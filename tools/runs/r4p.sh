mkdir -p gpurun_out/r4p
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 ) > gpurun_out/r4p/tests.log 2>&1
tail -3 gpurun_out/r4p/tests.log
bash tools/prof_pmc.sh r04 > gpurun_out/r4p/prof.log 2>&1
ls gpurun_out/r04/summary
( timeout 300 python bench.py --order-free --no-cpu-baseline --no-store-probe --pipeline-seconds 0 --e2e-seconds 0 2>/dev/null | python tools/ab_line.py ) > gpurun_out/r4p/order_free.log 2>&1
cat gpurun_out/r4p/order_free.log
( timeout 300 python bench.py --batch-reads 1000 --steps 200 --warmup 10 --no-cpu-baseline --no-store-probe --e2e-seconds 0 2> /dev/null ) > gpurun_out/r4p/bench_k1000.json
python -c "
import json; d=json.load(open('gpurun_out/r4p/bench_k1000.json')); p=d['pipeline']; print('K1000 value %.4e pipeline %.4e vs %.3f' % (d['value'], p['value'], p['vs_value']))"
