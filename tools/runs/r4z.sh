#!/bin/bash
# which launches of k_part_scan<64, 4> are slow (profiles/r04_kernel_stats.csv: 22 us .. 7.9 ms) and what runs around them
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r4z; rm -rf $OUT; mkdir -p $OUT
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-store-probe --pipeline-seconds 0 --e2e-seconds 1 > $OUT/t.log 2>&1
tail -2 $OUT/t.log
cd $R
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r4z/**/t_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
idx = [i for i, r in enumerate(rows) if "k_part_scan<64, 4>" in r["Kernel_Name"]]
print(len(idx), "launches of k_part_scan<64, 4>")
for i in idx[:3] + idx[len(idx) // 2: len(idx) // 2 + 3]:
    for r in rows[max(0, i - 3): i + 3]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"  {(s - t0) / 1e3:12.1f} us  +{(e - s) / 1e3:9.1f} us  q{r['Queue_Id']} grid {r['Grid_Size_X']}x{r['Grid_Size_Y']} wg {r['Workgroup_Size_X']}  {r['Kernel_Name'][:60]}")
    print()
d = [(int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3 for i in idx]
print("durations us:", [round(x) for x in d[:60]])
PY
