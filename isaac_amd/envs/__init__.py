"""Task registration (reference humanoid/envs/__init__.py:47-49): `hector` is the headline task, `hector_full` its
18-DoF sibling with actuated arms, `humanoid_ppo` the 12-DoF XBot-L humanoid the repository was derived from (same kernel
source, three instantiations)."""
from .configs import (HectorCfg, HectorCfgPPO, HectorFullCfg, HectorFullCfgPPO, LeggedRobotCfg,  # noqa: F401
                      LeggedRobotCfgPPO, XBotLCfg, XBotLCfgPPO)
from .hector_env import HectorFreeEnv, HectorFullFreeEnv, XBotLFreeEnv  # noqa: F401
from ..utils.task_registry import task_registry

task_registry.register("hector", HectorFreeEnv, HectorCfg(), HectorCfgPPO())
task_registry.register("hector_full", HectorFullFreeEnv, HectorFullCfg(), HectorFullCfgPPO())
task_registry.register("humanoid_ppo", XBotLFreeEnv, XBotLCfg(), XBotLCfgPPO())
