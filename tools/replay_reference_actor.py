#!/usr/bin/env python3
"""Container-only analysis: roll an actor the reference SHIPS (humanoid/locomotion_net*.onnx, trained against PhysX)
in this repository's physics model via the numpy oracle env (same model as the HIP simulator, parity-tested), and
report how long the robots stay up and how fast they walk.  Nothing is copied: the file is read in place.
usage: python tools/replay_reference_actor.py [path.onnx] [steps] [n_envs]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from isaac_amd.utils import onnx_io
from oracle.env import HectorEnvOracle

path = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/humanoid/locomotion_net.onnx"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
n = int(sys.argv[3]) if len(sys.argv) > 3 else 8
layers = onnx_io.load_actor(path)
rng = np.random.default_rng(0)
pack = lambda: np.concatenate([rng.uniform(size=(34, n)), rng.standard_normal((41, n))]).astype(np.float32)
origins = np.zeros((n, 3), np.float32)
origins[:, 0] = 3.0 * np.arange(n)
env = HectorEnvOracle(n, np.full(n, 0.8, np.float32), np.full(n, 8.15528, np.float32), origins, pack(), add_noise=False,
                      start_xy=origins.copy())
obs = env.obs_buf
alive = np.ones(n, bool)
first_fall = np.full(n, steps)
vx, height = [], []
for t in range(steps):
    a = onnx_io.mlp_forward(layers, obs)
    env.commands[:] = [0.5, 0.0, 0.0, 0.0]
    obs, priv, rew, done = env.step(a.astype(np.float32), pack())
    fell = done & ~env.time_out_buf
    first_fall = np.where(alive & fell, t, first_fall)
    alive &= ~fell
    vx.append(env.base_lin_vel[:, 0].copy())
    height.append(env.root[:, 2].copy())
vx, height = np.array(vx), np.array(height)
print(f"{os.path.basename(path)}: {n} robots, {steps} steps ({steps * 0.01:.1f} s), command vx = 0.5 m/s")
print("  first fall (step) per robot:", first_fall.tolist())
print("  robots never fallen: %d / %d" % (int((first_fall == steps).sum()), n))
print("  mean forward speed over steps 100.. of robots still up: %.3f m/s" % float(np.mean([vx[100:first_fall[i], i].mean() for i in range(n) if first_fall[i] > 150] or [np.nan])))
print("  mean base height while up: %.3f m" % float(np.mean([height[:first_fall[i], i].mean() for i in range(n) if first_fall[i] > 10] or [np.nan])))
