#!/bin/bash
# the whole GPU suite in the driver's form, N times on this box (five runs on three boxes): tools/runs6/suite2.sh <tag> [N]
cd "$(dirname "$0")/../.."
TAG=${1:-x}; N=${2:-2}
OUT=gpurun_out/suite_$TAG; mkdir -p $OUT
(hostname; rocm-smi --showserial 2>/dev/null | grep -i serial | head -2; python -c "from squigulator_amd import build; print('source_hash', build.source_hash())") > $OUT/box.txt 2>&1
for i in $(seq $N); do
  timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | sed "s/^/run $i: /" | tee -a $OUT/pytest.log
done
