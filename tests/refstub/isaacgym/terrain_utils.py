# not needed for mesh_type='plane' fixtures; present so `from isaacgym import terrain_utils` resolves
