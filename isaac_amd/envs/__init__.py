"""Task registration (reference humanoid/envs/__init__.py:47-49): `hector` is the headline task, `hector_full` its
18-DoF sibling with actuated arms (same kernel source, second instantiation)."""
from .configs import (HectorCfg, HectorCfgPPO, HectorFullCfg, HectorFullCfgPPO, LeggedRobotCfg,  # noqa: F401
                      LeggedRobotCfgPPO)
from .hector_env import HectorFreeEnv, HectorFullFreeEnv  # noqa: F401
from ..utils.task_registry import task_registry

task_registry.register("hector", HectorFreeEnv, HectorCfg(), HectorCfgPPO())
task_registry.register("hector_full", HectorFullFreeEnv, HectorFullCfg(), HectorFullCfgPPO())
