#!/usr/bin/env python3
"""Diagnostic (GPU box): train `iters` iterations on the default tile map, then roll the policy for `steps` steps and
report falls per robot-second by tile kind / difficulty of the tile each robot is assigned to."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from isaac_amd.envs import *  # noqa
from isaac_amd.utils import get_args, task_registry

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
args = get_args(["--task=hector", "--headless", "--run_name", "falls", "--max_iterations", str(iters)])
env_cfg, train_cfg = task_registry.get_cfgs("hector")
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    obj = env_cfg
    parts = k.split(".")
    for p in parts[:-1]:
        obj = getattr(obj, p)
    setattr(obj, parts[-1], eval(v))
env, _ = task_registry.make_env(name="hector", args=args, env_cfg=env_cfg)
runner, _ = task_registry.make_alg_runner(env=env, name="hector", args=args, log_root=None)
runner.learn(num_learning_iterations=iters, init_at_random_ep_len=True)
policy = runner.get_inference_policy()
n = env.num_envs
log = np.array(env.terrain.tile_log).reshape(env_cfg.terrain.num_rows, env_cfg.terrain.num_cols, 2)
kind = log[env.terrain_levels, env.terrain_types, 0].astype(int)
diff = log[env.terrain_levels, env.terrain_types, 1]
obs = env.get_observations()
falls = np.zeros(n)
touts = np.zeros(n)
for t in range(steps):
    obs, _, _, dones, infos = env.step(policy(obs))
    d = dones.numpy().astype(bool)
    to = env.time_out_buf.numpy().astype(bool)
    falls += d & ~to
    touts += to
names = ["flat", "obstacles", "rough", "slope up", "slope down", "stairs up", "stairs down"]
print(f"after {iters} iterations, {steps} evaluation steps ({steps * env.dt:.0f} s per robot)")
print("kind          robots  falls/robot/10s   by difficulty tercile (easy, mid, hard)")
for k in range(7):
    m = kind == k
    if not m.any():
        continue
    rate = lambda mm: falls[mm].sum() / max(mm.sum(), 1) / (steps * env.dt) * 10.0
    terc = [rate(m & (diff < 1 / 3)), rate(m & (diff >= 1 / 3) & (diff < 2 / 3)), rate(m & (diff >= 2 / 3))]
    print(f"{names[k]:12s} {m.sum():6d}  {rate(m):8.2f}          " + "  ".join(f"{x:6.2f}" for x in terc))
print("overall falls/robot/10s: %.2f" % (falls.sum() / n / (steps * env.dt) * 10))
