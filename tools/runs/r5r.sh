#!/bin/bash
cd "$(dirname "$0")/../.."
for args in "3088 1 0" "3088 0 0" "3088 1 1" "256 1 1"; do timeout 300 python tools/two_ctx_probe.py $args 2>&1 | tail -1; done
