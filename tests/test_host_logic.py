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


def _wgrad_plan(layers, rows, slots):
    import ctypes as C
    from isaac_amd import capi
    L = capi.lib()
    nl = len(layers)
    buf = (C.c_longlong * (12 * 16))()
    n, nlaunch, slab = C.c_int(0), C.c_int(0), C.c_longlong(0)
    capi.check(L.hx_wgrad_plan_describe(nl, (C.c_int * nl)(*[o for o, _ in layers]), (C.c_int * nl)(*[i for _, i in layers]), rows, slots, buf, 16,
                                        C.byref(n), C.byref(nlaunch), C.byref(slab)), "plan")
    keys = ("layer", "col0", "ncols", "shape", "bm", "bn", "tiles", "splits", "kchunk", "launch", "slab_off", "bslab_off")
    return [dict(zip(keys, buf[12 * k:12 * k + 12])) for k in range(n.value)], nlaunch.value, slab.value


def test_weight_gradient_plan_properties():
    """The host planner of the one-workgroup-per-CU weight gradients (isaac_amd/csrc/hx_wgrad_plan.h), no GPU needed: every column of
    every layer belongs to exactly one piece, the pieces of a launch fit the CUs, slices cover the rows in whole K tiles, slab regions
    do not overlap, and the workgroups of a launch are balanced (the point of the planner)."""
    import numpy as np
    hector = [(512, 616), (256, 512), (128, 256), (768, 1052), (256, 768), (128, 256)]          # actor then critic, as the learner orders them
    for layers, rows, slots in ((hector, 61440, 256), (hector, 61440, 304), (hector, 16384, 256), (hector, 960, 256),
                                ([(512, 708), (256, 512), (128, 256), (768, 220), (256, 768), (128, 256)], 61440, 256),
                                ([(100, 260), (36, 36), (132, 1060)], 1024, 40)):
        pieces, nlaunch, slab = _wgrad_plan(layers, rows, slots)
        assert 1 <= nlaunch <= 2 and 1 <= len(pieces) <= 12
        for l, (out, ind) in enumerate(layers):
            cover = np.zeros(ind, int)
            mine = [p for p in pieces if p["layer"] == l]
            for p in mine:
                cover[p["col0"]:p["col0"] + p["ncols"]] += 1
                assert p["col0"] % 4 == 0 and p["ncols"] % 4 == 0 and p["ncols"] > 0
                assert p["tiles"] == -(-out // p["bm"]) * -(-p["ncols"] // p["bn"])
            assert (cover == 1).all(), (l, cover)
            assert sum(1 for p in mine if p["bslab_off"] >= 0) == 1          # one piece per layer carries the bias gradient
        for q in range(nlaunch):
            wgs = sum(p["tiles"] * p["splits"] for p in pieces if p["launch"] == q)
            assert 0 < wgs <= slots, (q, wgs)
        for p in pieces:
            assert p["kchunk"] % 32 == 0 and p["splits"] * p["kchunk"] >= rows > (p["splits"] - 1) * p["kchunk"]
        spans = sorted((p["slab_off"], p["slab_off"] + p["splits"] * layers[p["layer"]][0] * p["ncols"]) for p in pieces)
        assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])) and spans[-1][1] == slab
    # the headline configuration: two launches of exactly one workgroup per CU, the critic's 1052-wide layer as 1024 + 28 columns, and
    # every workgroup's K loop (MFMA tiles per wave x rows per slice, the strip's weighted by its slower loop) within 12 % of the longest
    pieces, nlaunch, _ = _wgrad_plan(hector, 61440, 256)
    assert nlaunch == 2
    crit = sorted((p["col0"], p["ncols"], p["bm"], p["bn"]) for p in pieces if p["layer"] == 3)
    assert crit == [(0, 1024, 256, 256), (1024, 28, 512, 32)], crit
    units = {(256, 256): 16, (512, 128): 16 * 1.07, (128, 256): 8 * 1.10, (128, 128): 4 * 1.16, (512, 32): 4 * 1.30}
    for q in range(nlaunch):
        mine = [p for p in pieces if p["launch"] == q]
        assert sum(p["tiles"] * p["splits"] for p in mine) >= 250
        cost = [units[(p["bm"], p["bn"])] * (p["kchunk"] + 150) for p in mine]
        assert min(cost) > 0.88 * max(cost), (q, cost)
