mkdir -p gpurun_out/r4e
for abl in 1 2; do
cd /tmp && export TMPDIR=/tmp
SQG_PHC_ABL=$abl timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4e/abl$abl -o ks -- python $GRAFT_REPO_ROOT/bench.py --lib $GRAFT_REPO_ROOT/squigulator_amd/csrc/libsqg_hip_dev.so --steps 6 --warmup 2 --no-cpu-baseline --no-store-probe --pipeline-seconds 0 --e2e-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/r4e/abl$abl.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob
for f in glob.glob("gpurun_out/r4e/abl$abl/**/ks_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_part" in r["Name"] or "lean" in r["Name"]:
            print("abl$abl", f"{r['Name'][:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:10.1f} us")
PY
find gpurun_out/r4e -name "*.csv" -size +1M -delete
done
