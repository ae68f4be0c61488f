#!/usr/bin/env python3
"""HBM-side traffic of the GEMM kernels from the PMC counters.  The counters need their own passes (one counter each:
together they exceed what the hardware collects at once), run from the shell on the GPU box:

  cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
  for c in FETCH_SIZE WRITE_SIZE; do timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_traffic_<tag>/$c -o r -- \
      python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof; done
  python3 tools/collect_traffic.py <tag>

This script averages the counters per kernel and writes gpurun_out/<tag>_traffic.json (copy it to profiles/):
  fetch_bytes_per_launch = 2 x FETCH_SIZE (the gfx950 correction for wide coalesced reads, MI355X_MICROARCH.md "HBM")
  write_bytes_per_launch = WRITE_SIZE
bench.py reports these as roofline.traffic for the kernel it names (the counters cannot be read live there)."""
import collections, csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out_dir = os.path.join(ROOT, "gpurun_out", f"pmc_traffic_{tag}")
sys.path.insert(0, ROOT)
from isaac_amd import capi
BUILD_ID = capi.lib().hx_build_id().decode()          # the build this script runs beside = the build the passes ran on


def key_of(name):
    """rocprofv3's Kernel_Name -> the symbol as bench.py / include/hx_lab.h name it: 'void f<...>(Args)' -> 'f<...>'"""
    n = name[5:] if name.startswith("void ") else name
    depth = 0
    for i, ch in enumerate(n):
        depth += ch == "<"
        depth -= ch == ">"
        if ch == "(" and depth == 0:
            return n[:i]
    return n


agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        key = key_of(name)
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, c in agg.items():
    if "FETCH_SIZE" not in c:
        continue
    fetch_kb = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"])
    write_kb = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"]) if c.get("WRITE_SIZE") else 0.0
    res[k] = {"launches_sampled": len(c["FETCH_SIZE"]), "FETCH_SIZE_KB_raw": fetch_kb, "WRITE_SIZE_KB_raw": write_kb,
              "fetch_bytes_per_launch": 2.0 * fetch_kb * 1024.0, "write_bytes_per_launch": write_kb * 1024.0}
path = os.path.join(ROOT, "gpurun_out", f"{tag}_traffic.json")
json.dump({"build_id": BUILD_ID, "command": "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} (one pass each) -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof",
           "correction": "fetch = 2 x FETCH_SIZE (gfx950), write = WRITE_SIZE; units: rocprofv3 reports KB", "kernels": res}, open(path, "w"), indent=1)
print(path)
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["fetch_bytes_per_launch"])[:10]:
    print("%-70s fetch %8.1f MB  write %8.1f MB  (n=%d)" % (k[:70], v["fetch_bytes_per_launch"] / 1e6, v["write_bytes_per_launch"] / 1e6, v["launches_sampled"]))
