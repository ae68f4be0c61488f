#!/usr/bin/env python3
"""tests/golden/terrain_small.npz: height grids and tile origins produced by the REFERENCE's own
`HumanoidTerrain` (humanoid/utils/terrain.py:189-234 on Terrain :37-165) running over the oracle's restatement of
the `isaacgym.terrain_utils` primitives (tests/refstub/isaacgym/terrain_utils.py).  Pins the tile assembly: the
cumulative-proportion choice, difficulty scaling, tile placement, border and env-origin heights.
Two layouts: `rand_*` = randomized (np.random.seed(5), 4 x 7 tiles), `cur_*` = curriculum (3 x 10 tiles, every
tile kind).  Run in this container only."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests.refstub import loader  # noqa: E402

tmod = loader.load_terrain()
_, cfg_mod, _ = loader.load_env()


def small_cfg(rows, cols, curriculum, mesh_type):
    class T(cfg_mod.HectorCfg.terrain):
        pass
    T.num_rows, T.num_cols, T.curriculum, T.mesh_type, T.border_size = rows, cols, curriculum, mesh_type, 2.0
    return T


out = {}
np.random.seed(5)
t = tmod.HumanoidTerrain(small_cfg(4, 7, False, "trimesh"), 64)
out["rand_heights"], out["rand_origins"] = t.height_field_raw.copy(), t.env_origins.copy()
out["rand_vertices_sample"] = t.vertices[::97].copy()
out["rand_vertex_sum"] = t.vertices.astype(np.float64).sum(0)
out["rand_triangles_sample"] = t.triangles[::1013].copy()
np.random.seed(11)
t = tmod.HumanoidTerrain(small_cfg(3, 10, True, "heightfield"), 64)
out["cur_heights"], out["cur_origins"] = t.height_field_raw.copy(), t.env_origins.copy()
np.savez_compressed(os.path.join(HERE, "terrain_small.npz"), **out)
for k, v in out.items():
    print(k, v.shape, v.dtype)
print("kinds present (max |h| per tile, randomized):")
print(np.abs(out["rand_heights"][20:-20, 20:-20]).reshape(4, 80, 7, 80).max((1, 3)))
