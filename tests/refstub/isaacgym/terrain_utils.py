"""Stub of `isaacgym.terrain_utils` for running the reference's humanoid/utils/terrain.py in this container:
the names resolve to the oracle's restatement (oracle/terrain.py).  Test tooling only."""
from oracle.terrain import (SubTerrain, discrete_obstacles_terrain, pyramid_sloped_terrain,  # noqa: F401
                            pyramid_stairs_terrain, random_uniform_terrain, stepping_stones_terrain)
from oracle.terrain import heightfield_to_trimesh as convert_heightfield_to_trimesh  # noqa: F401
