mkdir -p gpurun_out/r4q
for wl in sequin-rna004 ncov-r9 synth-r10; do
( timeout 400 python bench.py --workload $wl --no-store-probe --cpu-seconds 3 2>/dev/null ) > gpurun_out/r4q/bench_$wl.json
python - <<PY
import json
d=json.load(open("gpurun_out/r4q/bench_$wl.json")); p=d["pipeline"]
print("$wl value %.4e ms/step %.3f lean %.3f events %.3f frac %.3f pipeline vs %.3f parity %s e2e blow5 %.3e" % (d["value"], d["ms_per_step"], d["kernel_ms"]["k_samples_lean"], d["kernel_ms"]["event side (k_events, k_part_*)"], d["roofline"]["frac"], p["vs_value"], d["parity_check"]["equal"], d["e2e"]["blow5"]["value"]))
PY
done
( timeout 300 python bench.py --workload ncov-r9 --workers-per-gpu 1 --no-cpu-baseline --no-store-probe --e2e-seconds 0 2>/dev/null | python tools/ab_line.py )
