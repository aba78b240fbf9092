#!/usr/bin/env python3
"""one-line digest of a bench.py JSON line on stdin (tools/ab_*.sh)"""
import json
import sys

d = None
for ln in sys.stdin:
    if ln.startswith('{"metric"'):
        d = json.loads(ln)
if d is None:
    print("no bench line")
else:
    km = d["kernel_ms"]
    print("lean %.3f ms  events %.3f ms  step %.3f ms  %.4e samples/s  fp64 fix-ups %.2e" % (
        km["k_samples_lean"], km["event side (k_events, k_part_*)"], d["ms_per_step"], d["value"], d.get("fp64_fixup_frac") or 0))
