"""CPU: host-side logic that needs no GPU -- creation-time randomisation in the reference's draw order,
config -> C struct derivation, checkpoint-format helpers, DeviceArray interface."""
import os

import numpy as np

from isaac_amd.envs.configs import HectorCfg
from isaac_amd.envs.hector_env import BASE_MASS, creation_randomisation
from isaac_amd.utils.helpers import get_args, set_seed, update_cfg_from_args

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_creation_randomisation_reproduces_reference_draws():
    """Same seed -> the reference's own per-env friction, payload and start pose (legged_robot.py:650-664),
    as recorded in the env fixture that the reference code generated with seed 5."""
    fx = np.load(os.path.join(GOLD, "env_rollout_a.npz"))
    n, _, seed, _, _ = (int(x) for x in fx["meta"])
    cfg = HectorCfg()
    cfg.env.num_envs = n
    set_seed(seed)
    fr, mass, start = creation_randomisation(cfg, n, fx["init_env_origins"])
    np.testing.assert_allclose(fr, fx["init_shape_friction"], rtol=1e-6)
    np.testing.assert_allclose(mass, fx["init_base_mass"], rtol=1e-6)
    np.testing.assert_allclose(start, fx["init_start_pos"], rtol=0, atol=1e-6)
    assert np.all(mass >= BASE_MASS - 2) and np.all(mass <= BASE_MASS + 4)


def test_get_args_and_overrides():
    a = get_args(["--task=hector", "--headless", "--num_envs", "64", "--seed", "9", "--max_iterations", "3", "--run_name", "v1"])
    assert a.task == "hector" and a.headless and a.num_envs == 64 and a.sim_device == "cuda:0"
    from isaac_amd.envs.configs import HectorCfgPPO
    env_cfg, tr = update_cfg_from_args(HectorCfg(), HectorCfgPPO(), a)
    assert env_cfg.env.num_envs == 64 and tr.seed == 9 and tr.runner.max_iterations == 3 and tr.runner.run_name == "v1"


def test_registry_surface():
    import pytest
    from isaac_amd.envs import task_registry
    env_cfg, train_cfg = task_registry.get_cfgs("hector")
    assert env_cfg.seed == train_cfg.seed == 5                       # task_registry.py:62 copies the seed
    with pytest.raises(ValueError, match="was not registered"):
        task_registry.make_env("nope", args=get_args([]))


def test_actor_critic_state_dict_layout():
    from isaac_amd.algo.ppo import ActorCritic
    ac = ActorCritic(615, 1050, 10, [512, 256, 128], [768, 256, 128])
    keys = list(ac.state_dict())
    assert keys[0] == "std" and keys[1:3] == ["actor.0.weight", "actor.0.bias"] and keys[-1] == "critic.6.bias"
    assert ac.num_params() == 1_517_973
    sd = ac.state_dict()
    assert sd["actor.0.weight"].shape == (512, 615) and abs(sd["actor.0.weight"]).max() <= 1 / np.sqrt(615) + 1e-6
