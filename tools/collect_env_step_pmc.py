#!/usr/bin/env python3
"""Wave-level instruction counters of the env-step kernel from a PMC pass.  Run from the shell on the GPU box:

  cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d gpurun_out/pmc_env_step_<tag> -o r -- \
      python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof
  python3 tools/collect_env_step_pmc.py <tag>

Writes gpurun_out/<tag>_env_step_pmc.json (copy it to profiles/): per launch of hx_env_step_kernel the mean of every
counter; bench.py divides valu_insts_per_launch by the live HIP-event duration of the kernel for roofline.env_step."""
import collections, csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
out_dir = os.path.join(ROOT, "gpurun_out", f"pmc_env_step_{tag}")
agg = collections.defaultdict(list)
for f in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "hx_env_step_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
assert agg, "no hx_env_step_kernel rows under " + out_dir
mean = {k: sum(v) / len(v) for k, v in agg.items()}
sys.path.insert(0, ROOT)
from isaac_amd import capi
res = {"build_id": capi.lib().hx_build_id().decode(), "command": "rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof",
       "kernel": "hx_env_step_kernel<ModelHector>", "terrain": "trimesh", "envs": 4096, "launches_sampled": len(next(iter(agg.values()))),
       "valu_insts_per_launch": mean.get("SQ_INSTS_VALU"), "waves_per_launch": mean.get("SQ_WAVES"),
       "salu_insts_per_launch": mean.get("SQ_INSTS_SALU"), "lds_insts_per_launch": mean.get("SQ_INSTS_LDS")}
if res["waves_per_launch"]:
    res["valu_insts_per_wave"] = res["valu_insts_per_launch"] / res["waves_per_launch"]
path = os.path.join(ROOT, "gpurun_out", f"{tag}_env_step_pmc.json")
json.dump(res, open(path, "w"), indent=1)
print(path, json.dumps(res))
