#!/usr/bin/env python3
"""Per-step composition of the rollout from a `rocprofv3 --kernel-trace --output-format csv` run of bench.py (row storage): durations of the fused
actor, the env step and the stacking kernel, the gaps between them and the step period.  usage: rollout_timeline.py <output dir>"""
import csv, glob, os, statistics, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    k = "actor" if "actor_fused" in n else "env" if "env_step" in n else "stack" if "hx_stack_kernel" in n else None
    if k:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k))
rows.sort()
seq = []
for i in range(len(rows) - 3):
    a, e, s, nx = rows[i:i + 4]
    if (a[2], e[2], s[2], nx[2]) == ("actor", "env", "stack", "actor"):
        seq.append(dict(actor=(a[1] - a[0]) / 1e3, gap_actor_env=(e[0] - a[1]) / 1e3, env=(e[1] - e[0]) / 1e3, gap_env_stack=(s[0] - e[1]) / 1e3,
                        stack=(s[1] - s[0]) / 1e3, gap_stack_actor=(nx[0] - s[1]) / 1e3, step=(nx[0] - a[0]) / 1e3))
seq = [x for x in seq if x["step"] < 1000]            # drop the steps that span an update
print(f"{len(seq)} rollout steps")
for k in seq[0]:
    v = sorted(x[k] for x in seq)
    print(f"  {k:16s} median {statistics.median(v):7.1f}  mean {statistics.mean(v):7.1f}  p10 {v[len(v) // 10]:7.1f}  p90 {v[9 * len(v) // 10]:7.1f} us")
