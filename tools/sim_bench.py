#!/usr/bin/env python3
"""Env-step-only timing: `steps` hx_sim_step calls on N robots, perf-mode RNG.  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from isaac_amd import capi
from isaac_amd.envs.configs import HectorCfg, HectorFullCfg
from isaac_amd.envs.hector_env import HectorFreeEnv, HectorFullFreeEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
mesh = sys.argv[3] if len(sys.argv) > 3 else "trimesh"      # trimesh | heightfield (no walls) | plane | flatgrid (terrain path over an all-zero grid)
task = sys.argv[4] if len(sys.argv) > 4 else "hector"          # hector | hector_full
sigma = float(sys.argv[5]) if len(sys.argv) > 5 else 0.3       # standard deviation of the (fixed) random actions: 0 = robots stand, nobody falls
cfg = (HectorFullCfg if task == "hector_full" else HectorCfg)(); cfg.env.num_envs = n; cfg.seed = 5
cfg.terrain.mesh_type = "plane" if mesh == "plane" else ("heightfield" if mesh == "heightfield" else "trimesh")
if mesh == "flatgrid":
    cfg.terrain.terrain_proportions = [1.0, 0, 0, 0, 0, 0, 0]
np.random.seed(5)
env = (HectorFullFreeEnv if task == "hector_full" else HectorFreeEnv)(cfg)
act = capi.DeviceBuffer.from_host((sigma * np.random.default_rng(0).standard_normal((n, env.num_actions))).astype(np.float32))
L = capi.lib()
for _ in range(20):
    L.hx_sim_step(env._h, act.ptr, None)
env.sync()
t0 = time.perf_counter()
for _ in range(steps):
    L.hx_sim_step(env._h, act.ptr, None)
env.sync()
dt = (time.perf_counter() - t0) / steps
ep = env._buf(capi.BUF_EP_LEN, (n,), np.int32).numpy()
print(f"actions sigma {sigma}: mean episode length so far {ep.mean():.0f} steps")
print(f"N={n} {task} {mesh}: {dt*1e6:.1f} us per env step (incl. 2 stack kernels + memset), {n/dt/1e6:.2f} M env-steps/s sim-only")
