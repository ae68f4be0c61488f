#!/usr/bin/env python3
"""Per-kernel MFMA pipe utilisation from a `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
--output-format csv` run.  usage: mfma_util_from_pmc.py <counter_collection.csv> <kernel_trace.csv> <out.json>"""
import collections, csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isaac_amd import build as _b
cc, kt, out_path = sys.argv[1:4]
BUILD_ID = "100-" + _b.source_hash()          # the sources this script runs beside = the build the pass ran on (hx_build_id)
tr = {r["Dispatch_Id"]: r for r in csv.DictReader(open(kt))}
agg = collections.defaultdict(lambda: dict(busy=0.0, dur=0.0, n=0, gui=0.0))
for r in csv.DictReader(open(cc)):
    t = tr.get(r["Dispatch_Id"])
    if not t:
        continue
    a = agg[r["Kernel_Name"].split("(")[0].replace("void ", "")]
    if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
        a["busy"] += float(r["Counter_Value"]); a["dur"] += int(t["End_Timestamp"]) - int(t["Start_Timestamp"]); a["n"] += 1
    elif r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        a["gui"] += float(r["Counter_Value"])
out = dict(build_id=BUILD_ID, command="rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof",
           note="mfma_pipe_util = SQ_VALU_MFMA_BUSY_CYCLES / (kernel duration x 2.4 GHz x 1024 SIMDs); durations under counter collection are longer than in the plain kernel trace",
           kernels={})
for k, a in sorted(agg.items(), key=lambda x: -x[1]["dur"]):
    if a["busy"] == 0 or a["n"] == 0:
        continue
    out["kernels"][k] = dict(launches=a["n"], mfma_busy_cycles_per_launch=a["busy"] / a["n"], avg_us=a["dur"] / a["n"] / 1e3,
                             grbm_gui_active_per_launch=a["gui"] / a["n"], mfma_pipe_util_at_2p4GHz=round(a["busy"] / (a["dur"] * 2.4 * 1024), 3),
                             effective_clock_ghz=round(a["gui"] / 8.0 / a["dur"], 3) if a["gui"] else None)
    print(k, a["n"], round(a["dur"] / a["n"] / 1e3, 1), out["kernels"][k]["mfma_pipe_util_at_2p4GHz"])
json.dump(out, open(out_path, "w"), indent=1)
