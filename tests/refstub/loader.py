"""Import the reference's Python modules in THIS container without running its package __init__s
(which pull in isaacgym / wandb / tensorboard).  Test tooling only; requires /root/reference."""
import importlib
import os
import sys
import types

REF = os.environ.get("HX_REFERENCE_ROOT", "/root/reference")
_HERE = os.path.dirname(os.path.abspath(__file__))
_REPO = os.path.dirname(os.path.dirname(_HERE))


def available():
    return os.path.isdir(os.path.join(REF, "humanoid"))


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


def load_ppo():
    """reference humanoid/algo/ppo/{actor_critic,rollout_storage,ppo}.py as package `refppo`."""
    sys.dont_write_bytecode = True
    if "refppo" not in sys.modules:
        _pkg("refppo", os.path.join(REF, "humanoid/algo/ppo"))
    ac = importlib.import_module("refppo.actor_critic")
    rs = importlib.import_module("refppo.rollout_storage")
    ppo = importlib.import_module("refppo.ppo")
    return ac, rs, ppo


def load_env(task="hector"):
    """reference HectorFreeEnv / HectorCfg (task "hector"), HectorFullFreeEnv / HectorFullCfg ("hector_full") or XBotLFreeEnv /
    XBotLCfg ("humanoid_ppo") over the stub isaacgym (tests/refstub/isaacgym)."""
    sys.dont_write_bytecode = True
    if _REPO not in sys.path:
        sys.path.insert(0, _REPO)
    if _HERE not in sys.path:
        sys.path.insert(0, _HERE)          # makes `import isaacgym` resolve to the stub
    import isaacgym  # noqa: F401  (the stub)
    if "humanoid" not in sys.modules:
        h = _pkg("humanoid", os.path.join(REF, "humanoid"))
        h.LEGGED_GYM_ROOT_DIR = REF
        h.LEGGED_GYM_ENVS_DIR = os.path.join(REF, "humanoid", "envs")
        _pkg("humanoid.utils", os.path.join(REF, "humanoid/utils"))
        e = _pkg("humanoid.envs", os.path.join(REF, "humanoid/envs"))
        _pkg("humanoid.envs.base", os.path.join(REF, "humanoid/envs/base"))
        _pkg("humanoid.envs.custom", os.path.join(REF, "humanoid/envs/custom"))
        lr = importlib.import_module("humanoid.envs.base.legged_robot")
        e.LeggedRobot = lr.LeggedRobot
    stem = {"hector_full": "hector_w_arm", "humanoid_ppo": "humanoid"}.get(task, "hector")
    env_mod = importlib.import_module(f"humanoid.envs.custom.{stem}_env")
    cfg_mod = importlib.import_module(f"humanoid.envs.custom.{stem}_config")
    helpers = importlib.import_module("humanoid.utils.helpers")
    return env_mod, cfg_mod, helpers


def load_terrain():
    """reference humanoid/utils/terrain.py (Terrain, HumanoidTerrain) over the stub `isaacgym.terrain_utils`."""
    load_env()
    return importlib.import_module("humanoid.utils.terrain")
