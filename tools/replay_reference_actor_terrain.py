#!/usr/bin/env python3
"""Container-only analysis: the actor the reference ships (trained against PhysX on its tile map) rolled zero-shot on THIS
repository's terrain contact model (numpy oracle env + oracle/terrain.py), one robot per tile of a curriculum-layout map
(every tile kind at three difficulties).  Reports survival and forward speed per tile kind.
usage: python tools/replay_reference_actor_terrain.py [path.onnx] [steps]"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from isaac_amd.utils import onnx_io
from oracle.env import HectorEnvOracle
from oracle.terrain import HeightField, HumanoidTerrainOracle

path = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/humanoid/locomotion_net.onnx"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
layers = onnx_io.load_actor(path)
rows, cols = 3, 10
tc = types.SimpleNamespace(mesh_type="trimesh", horizontal_scale=0.1, vertical_scale=0.005, border_size=5.0, curriculum=True,
                           selected=False, terrain_length=8.0, terrain_width=8.0, num_rows=rows, num_cols=cols,
                           terrain_proportions=[0.1, 0.1, 0.2, 0.1, 0.1, 0.2, 0.2])
np.random.seed(11)
ter = HumanoidTerrainOracle(tc, rows * cols)
hf = HeightField(ter.heightsamples, 0.1, 0.005, 5.0)
kinds = ["flat", "obstacles", "rough", "rough", "slope up", "slope down", "stairs up", "stairs up", "stairs down", "stairs down"]
n = rows * cols
origins = ter.env_origins.reshape(n, 3).astype(np.float32)          # row-major: (level i, type j)
rng = np.random.default_rng(0)
pack = lambda: np.concatenate([rng.uniform(size=(34, n)), rng.standard_normal((41, n))]).astype(np.float32)
p0 = pack(); p0[29:31] = 0.5
env = HectorEnvOracle(n, np.full(n, 0.8, np.float32), np.full(n, 8.15528, np.float32), origins, p0, add_noise=False,
                      start_xy=origins.copy(), terrain=hf, custom_origins=True)
obs = env.obs_buf
first_fall = np.full(n, steps)
alive = np.ones(n, bool)
x0 = env.root[:, 0].copy()
dist = np.zeros(n)
for t in range(steps):
    a = onnx_io.mlp_forward(layers, obs)
    env.commands[:] = [0.5, 0.0, 0.0, 0.0]
    pk = pack(); pk[29:31] = 0.5
    obs, priv, rew, done = env.step(a.astype(np.float32), pk)
    fell = done & ~env.time_out_buf
    first_fall = np.where(alive & fell, t, first_fall)
    dist = np.where(alive & ~fell, env.root[:, 0] - x0, dist)
    alive &= ~fell
print(f"{os.path.basename(path)} on the tile map, {steps} steps ({steps * 0.01:.1f} s), command vx = 0.5 m/s, one robot per tile")
print("difficulty   " + "  ".join(f"{k:>11s}" for k in kinds))
for i in range(rows):
    cells = []
    for j in range(cols):
        e = i * cols + j
        cells.append(f"{first_fall[e]:4d}/{dist[e]:5.2f}m")
    print(f"   {i / rows:4.2f}      " + "  ".join(f"{c:>11s}" for c in cells))
print("(cell = first fall step / distance walked along x while up; %d = never fell)" % steps)
