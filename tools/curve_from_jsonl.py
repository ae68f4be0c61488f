#!/usr/bin/env python3
"""Print a training curve (every <stride> iterations) from the runner's scalars.jsonl.
usage: curve_from_jsonl.py <scalars.jsonl> [stride] [extra scalar names ...]"""
import json, sys
path, stride = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 100
keys = ["Train/mean_reward", "Train/mean_episode_length", "Episode/rew_tracking_lin_vel", "Loss/learning_rate",
        "Policy/mean_noise_std", "Perf/total_fps"] + sys.argv[3:]
print("it  " + "  ".join(keys))
rows = {}
for line in open(path):
    r = json.loads(line)
    it = r.get("it", r.get("iteration", r.get("step")))
    rows.setdefault(it, {}).update(r)
for it in sorted(rows):
    if it % stride == 0 or it == max(rows):
        r = rows[it]
        print("%5d " % it + "  ".join(("%10.4g" % r[k]) if k in r else "       n/a" for k in keys))
