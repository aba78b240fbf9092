#!/bin/bash
# round 6's record: the driver's default bench line, the other two workloads, then the rocprofv3 passes (kernel trace + stats, PMC groups) -> gpurun_out/r06
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=gpurun_out/r06; mkdir -p $OUT
(rocm-smi --showserial 2>/dev/null | grep -i serial | head -1; python -c "from squigulator_amd import build; print('source_hash', build.source_hash())") > $OUT/box.txt 2>&1
[ -n "$PROF_ONLY" ] || timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
[ -n "$PROF_ONLY" ] || timeout 600 python bench.py --workload ncov-r9 > $OUT/bench_ncov_r9.json 2> $OUT/bench_ncov_r9.err
[ -n "$PROF_ONLY" ] || timeout 600 python bench.py --workload sequin-rna004 > $OUT/bench_rna004.json 2> $OUT/bench_rna004.err
timeout 2400 bash tools/prof_pmc.sh r06prof > $OUT/prof.log 2>&1
cp -r gpurun_out/r06prof/summary $OUT/ 2>/dev/null
tail -c 600 $OUT/bench.json | head -c 600; echo; ls $OUT $OUT/summary
