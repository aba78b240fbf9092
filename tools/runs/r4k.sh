python tools/k1000_probe.py 1000 400
python tools/k1000_probe.py 1000 400
