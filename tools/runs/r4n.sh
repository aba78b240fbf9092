( timeout 900 python -m pytest tests/test_sampler.py tests/test_config2_hg38.py tests/test_precount.py -m gpu -q -x 2>&1 | tail -4 )
python tools/k1000_probe.py 1000 400; python tools/k1000_probe.py 1000 400
SQG_STAGE_TIMING=1 python tools/k1000_probe.py 1000 30 2>&1 | grep "^\[stage\]" | tail -7
