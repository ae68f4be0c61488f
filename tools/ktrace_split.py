#!/usr/bin/env python3
"""Durations of one kernel symbol in a `rocprofv3 --kernel-trace --output-format csv` run, split by launch position in a repeating
pattern (e.g. the two alternating shapes of hx_wgrad_multi_kernel).  usage: ktrace_split.py <output dir> <name substring> [period=2]"""
import csv, glob, os, statistics, sys
d, sub = sys.argv[1], sys.argv[2]; period = int(sys.argv[3]) if len(sys.argv) > 3 else 2
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size", "?")))
rows.sort()
for p in range(period):
    v = [r[1] for r in rows[p::period]]
    if v: print(f"{sub} position {p} of {period}: {len(v)} launches, grid {rows[p][2]}, median {statistics.median(v):.1f} us, mean {statistics.mean(v):.1f}, min {min(v):.1f}, max {max(v):.1f}")
