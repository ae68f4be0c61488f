"""VecEnv protocol; defined in isaac_amd/algo/vec_env.py (where the reference keeps it: humanoid/algo/vec_env.py)."""
from ..algo.vec_env import VecEnv  # noqa: F401
