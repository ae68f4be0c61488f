"""Task registration (reference humanoid/envs/__init__.py:47-49): `hector` is the task this build serves."""
from .configs import HectorCfg, HectorCfgPPO, LeggedRobotCfg, LeggedRobotCfgPPO  # noqa: F401
from .hector_env import HectorFreeEnv  # noqa: F401
from ..utils.task_registry import task_registry

task_registry.register("hector", HectorFreeEnv, HectorCfg(), HectorCfgPPO())
