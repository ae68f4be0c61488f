#!/usr/bin/env python3
"""Mean per launch of every counter of one kernel over all rocprofv3 counter_collection CSVs under the given directories.
usage: python tools/pmc_mean.py <kernel-name-substring> <dir> [<dir> ...]"""
import collections, csv, glob, os, sys
pat = sys.argv[1]
agg = collections.defaultdict(list)
for d in sys.argv[2:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                agg["_duration_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for k in sorted(agg):
    v = agg[k]
    print(f"{k:32s} n={len(v):5d} mean={sum(v) / len(v):16.1f}")
