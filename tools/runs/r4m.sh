SQG_STAGE_TIMING=1 python tools/k1000_probe.py 1000 30 2>&1 | grep "^\[stage\]" | tail -21
SQG_NO_DRAW_AHEAD=1 python tools/k1000_probe.py 1000 400
python tools/k1000_probe.py 1000 400
