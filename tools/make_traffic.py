#!/usr/bin/env python3
"""HBM traffic per launch from the FETCH_SIZE / WRITE_SIZE passes of tools/prof_pmc.sh.
usage: python tools/make_traffic.py gpurun_out/r01 profiles/r01 [workload_key]
Writes <dst>_traffic.json and profiles/traffic_latest.json (read by bench.py's roofline.traffic).
Units: both counters are KiB; FETCH_SIZE is doubled (gfx950 correction, MI355X_MICROARCH.md, HBM section);
WRITE_SIZE is calibrated in the same pass on k_store_probe, which writes exactly 1 GiB per launch."""
import collections
import csv
import json
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
key = sys.argv[3] if len(sys.argv) > 3 else None
if key is None:                                            # the workload key of the profiled bench line
    for ln in open(os.path.join(src, "trace.log")):
        if ln.startswith('{"metric"'):
            key = json.loads(ln)["roofline"]["workload_key"]
_m = __import__("re").search(r"r(\d+)$", os.path.basename(dst))
rnd = int(sys.argv[4]) if len(sys.argv) > 4 else int(_m.group(1)) if _m else None          # profiles/r03 -> 3
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f, name in (("pmc4", "FETCH_SIZE"), ("pmc5", "WRITE_SIZE")):
    path = os.path.join(src, f + "_counter_collection.csv")
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            acc[r["Kernel_Name"]][name].append(float(r["Counter_Value"]))
# round 5: what the kernels occupy besides bytes -- VALU issue (pmc1), the launch's cycles (pmc3), TCP -> L2 requests (pmc6), and the
# launch durations of the pmc3 pass (the clock the cycles were counted at)
def _short(kn):
    short = kn.split("(")[0].split("<")[0].replace("void ", "").strip()
    if short in ("k_events", "k_part_events"):
        short = kn.split("(")[0].replace("void ", "").strip()
    return short
extra = collections.defaultdict(lambda: collections.defaultdict(list))
for f, names in (("pmc1", ("SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES")), ("pmc3", ("GRBM_GUI_ACTIVE",)),
                 ("pmc6", ("TCP_TCC_READ_REQ_sum", "TCP_TCC_WRITE_REQ_sum"))):
    path = os.path.join(src, f + "_counter_collection.csv")
    if not os.path.exists(path):
        continue
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] in names:
            extra[_short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
path = os.path.join(src, "pmc3_kernel_trace.csv")
if os.path.exists(path):
    for r in csv.DictReader(open(path)):
        dur[_short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
calib = {}
path = os.path.join(src, "cal1_counter_collection.csv")
if os.path.exists(path):
    cc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if "cal_rgather8" in r["Kernel_Name"]:
            cc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    if cc.get("TCP_TCC_READ_REQ_sum") and cc.get("GRBM_GUI_ACTIVE"):
        rq = sum(cc["TCP_TCC_READ_REQ_sum"]) / len(cc["TCP_TCC_READ_REQ_sum"]); cy = sum(cc["GRBM_GUI_ACTIVE"]) / len(cc["GRBM_GUI_ACTIVE"]) / 8
        calib = {"rgather8_l2_req_per_cycle": rq / cy, "rgather8_requests": rq, "rgather8_cycles": cy,
                 "what": "tools/pmc_calib.hip cal_rgather8: 8 B per lane at random places of a 2-MiB (L2-resident) table; TCP_TCC_READ_REQ_sum over GRBM_GUI_ACTIVE / 8"}
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from squigulator_amd import build as _build  # noqa: E402
out = {"round": rnd, "workload_key": key, "source_hash": _build.source_hash(),
       "method": "per launch, averaged over the launches of the pass.  SQ_* (pmc1), GRBM_GUI_ACTIVE + kernel_us (pmc3), TCP_TCC_*_REQ (pmc6) as counted; "
                 "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/prof_pmc.sh); unit KiB; "
                 "FETCH_SIZE doubled per the gfx950 correction (MI355X_MICROARCH.md, HBM); WRITE_SIZE checked on "
                 "k_store_probe (1 GiB written per launch)",
       "kernels": {}}
for kn, d in acc.items():
    short = kn.split("(")[0].split("<")[0].replace("void ", "").strip()
    if not short.startswith("k_"):
        continue
    if short in ("k_events", "k_part_events"):             # several instantiations per batch (count / scatter passes): keep them apart
        short = kn.split("(")[0].replace("void ", "").strip()
    f = sum(d.get("FETCH_SIZE", [0])) / max(len(d.get("FETCH_SIZE", [0])), 1)
    w = sum(d.get("WRITE_SIZE", [0])) / max(len(d.get("WRITE_SIZE", [0])), 1)
    out["kernels"][short] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "hbm_bytes_per_launch": (2 * f + w) * 1024}
    for cn, v in extra.get(short, {}).items():
        out["kernels"][short][cn] = sum(v) / len(v)
    if dur.get(short):
        out["kernels"][short]["kernel_us"] = sum(dur[short]) / len(dur[short])
out["calib"] = calib
if True:
    pass
for path in (dst + "_traffic.json", os.path.join(os.path.dirname(dst) or ".", "traffic_latest.json")):
    json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
