#!/bin/bash
# k_part_scan keeping a run's counts in registers between its two loops (SCAN_KEEP 8 / 4) against reading every cell twice (0): the kernel's own duration
# under rocprofv3 --kernel-trace, then the whole step
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5ad; mkdir -p $OUT
export TMPDIR=/tmp
SQG_LIB=$PWD/tools/var_b_keep8.so timeout 600 python -m pytest tests/test_fuzz_parity.py tests/test_split_chains.py tests/test_range_sharding.py tests/test_many_reads.py -m gpu -x -q 2>&1 | tail -3 | tee $OUT/pytest.log
for rep in 1 2; do
for f in tools/var_*.so; do
  n=$(basename $f .so)
  rm -rf $OUT/kt_$n
  timeout 600 rocprofv3 --kernel-trace -d $OUT/kt_$n -o kt --output-format csv -- python bench.py --lib $PWD/$f --no-cpu-baseline --no-store-probe --pipeline-seconds 0 --e2e-seconds 0 --small-batch-seconds 0 --every-batch-launches 0 --steps 12 > $OUT/bench_$n.log 2>&1
  python - $OUT/kt_$n $n <<'PY'
import csv, glob, sys, statistics
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
d = {}
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0]
    d.setdefault(k, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k in sorted(d):
    if 'k_part_scan' in k or 'k_part_hist' in k or 'k_part_hand_count' in k:
        big = [x for x in d[k] if x > 0.5 * max(d[k])]
        print(f"{sys.argv[2]:16s} {k[:40]:40s} n {len(big):3d} median {statistics.median(big):7.1f} us")
PY
done
done 2>&1 | tee $OUT/scan.log
REPS=2 bash tools/ab_step.sh 2>&1 | tee $OUT/ab.log
