#!/usr/bin/env python3
"""Per-kernel summary of a `rocprofv3 --kernel-trace --stats --output-format csv` run: calls, total ms, average us, share.
usage: kstats.py <output dir> [top N]   (reads *kernel_stats.csv, else aggregates *kernel_trace.csv)"""
import collections, csv, glob, os, sys
d = sys.argv[1]; top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows = []
st = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
if st:
    for r in csv.DictReader(open(st[0])):
        rows.append((r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6))
else:
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            a = agg[r["Kernel_Name"]]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    rows = [(k, v[0], v[1]) for k, v in agg.items()]
tot = sum(r[2] for r in rows)
print("%-110s %7s %10s %9s %6s" % ("kernel", "calls", "total ms", "avg us", "%"))
for n, c, ms in sorted(rows, key=lambda r: -r[2])[:top]:
    print("%-110s %7d %10.2f %9.1f %6.1f" % (n.replace("void ", "")[:110], c, ms, 1e3 * ms / c, 100 * ms / tot))
print("total %.1f ms over %d kernels" % (tot, len(rows)))
