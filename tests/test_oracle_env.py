"""CPU: the env-glue oracle (oracle/env.py) replays the golden trajectories produced by the reference's own
HectorFreeEnv (tests/golden/make_env_fixtures.py) and must reproduce them step for step.
This is what pins the oracle to the reference for SURVEY.md 8a rows a1-a3, a5-a11."""
import os

import numpy as np
import pytest

from oracle.env import REWARD_ORDER, REWARD_SCALE, HectorEnvOracle

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def make_oracle_env(fx, **kw):
    """HectorEnvOracle seeded with a fixture's creation data (terrain included when the fixture has one)."""
    n, _, _, _, noise = (int(x) for x in fx["meta"])
    terrain = None
    if "terrain_heights" in fx:
        from oracle.terrain import HeightField
        hs, vs, border = fx["terrain_params"]
        terrain = HeightField(fx["terrain_heights"], hs, vs, border, wall_height=float(fx["terrain_wall_height"]) if "terrain_wall_height" in fx else 0.0)
    if "task" in fx and str(fx["task"]) != "hector":
        from oracle.env import HECTOR_FULL, HUMANOID
        kw.setdefault("task", {"hector_full": HECTOR_FULL, "humanoid_ppo": HUMANOID}[str(fx["task"])])
    if "cfg_override_names" in fx and len(fx["cfg_override_names"]):
        import json
        ov = {str(k): json.loads(str(v)) for k, v in zip(fx["cfg_override_names"], fx["cfg_override_values"])}
        opts = {"cmd_ranges": {}}
        for k, v in ov.items():
            if k.startswith("commands.ranges."):
                opts["cmd_ranges"][k.split(".")[-1]] = tuple(v)
            elif k in ("commands.heading_command", "rewards.only_positive_rewards", "domain_rand.push_robots"):
                opts[k.split(".")[-1]] = bool(v)
            elif k in ("domain_rand.action_delay", "domain_rand.action_noise"):
                opts[k.split(".")[-1]] = float(v)
            else:
                raise AssertionError(f"fixture override {k} has no oracle counterpart")
        kw.setdefault("opts", opts)
    if "reward_override_names" in fx and len(fx["reward_override_names"]):
        kw.setdefault("reward_scales", {str(k): float(v) for k, v in zip(fx["reward_override_names"], fx["reward_override_values"])})
    if "terrain_curriculum" in fx and int(fx["terrain_curriculum"]):
        kw.setdefault("curriculum", dict(origins=fx["terrain_origins"], levels=fx["init_terrain_levels"], types=fx["terrain_types"],
                                         env_length=float(fx["terrain_env_length"])))
    return HectorEnvOracle(n, fx["init_shape_friction"], fx["init_base_mass"], fx["init_env_origins"], fx["packs"][0],
                           add_noise=bool(noise), start_xy=fx["init_start_pos"], terrain=terrain,
                           custom_origins=terrain is not None, **kw)


@pytest.mark.parametrize("name,steps", [("env_rollout_a", 40), ("env_rollout_b", 40), ("env_rollout_c", 30), ("env_rollout_d", 150),
                                        ("env_rollout_e", 90), ("env_rollout_f", 60), ("env_rollout_g", 80), ("env_rollout_h", 80)])
def test_oracle_env_reproduces_reference(name, steps):
    fx = np.load(os.path.join(GOLD, name + ".npz"))
    n, total, seed, sc0, noise = (int(x) for x in fx["meta"])
    env = make_oracle_env(fx)
    np.testing.assert_allclose(env.obs_buf, fx["init_obs_full"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(env.priv_buf, fx["init_priv_full"], rtol=0, atol=1e-6)
    assert list(fx["reward_names"]) == env.reward_order      # alphabetical dir() order (helpers.py:47)
    if name not in ("env_rollout_e", "env_rollout_g", "env_rollout_h"):
        assert env.reward_order == REWARD_ORDER
    if name == "env_rollout_f":
        assert fx["rew"].min() < 0 and not env.opts["heading_command"]       # unclipped rewards, yaw-rate commands
    np.testing.assert_allclose(fx["reward_scales"], [env.reward_scale[k] * 0.01 for k in env.reward_order], rtol=1e-12)
    env.episode_length_buf[:] = fx["ep_len_init"]
    env.common_step_counter = sc0
    full = {int(s): i for i, s in enumerate(fx["full_steps"])}
    for t in range(min(steps, total)):
        obs, priv, rew, reset = env.step(fx["actions"][t], fx["packs"][t + 1])
        np.testing.assert_allclose(obs[:, -env.task.nobs:], fx["obs41"][t], rtol=0, atol=1e-4, err_msg=f"obs step {t}")
        np.testing.assert_allclose(priv[:, -env.task.npriv:], fx["priv70"][t], rtol=0, atol=1e-4, err_msg=f"priv step {t}")
        np.testing.assert_allclose(rew, fx["rew"][t], rtol=0, atol=1e-6)
        assert np.array_equal(reset, fx["reset"][t].astype(bool))
        assert np.array_equal(env.time_out_buf, fx["timeout"][t].astype(bool))
        assert np.array_equal(env.time_outs_visible, fx["timeouts_visible"][t].astype(bool))     # stale-extras quirk
        np.testing.assert_allclose(env.torques, fx["torques"][t], rtol=0, atol=2e-3)
        np.testing.assert_allclose(env.commands, fx["commands"][t], rtol=0, atol=1e-5)
        np.testing.assert_array_equal(env.episode_length_buf, fx["ep_len"][t])
        np.testing.assert_allclose(env.feet_air_time, fx["feet_air_time"][t], rtol=0, atol=1e-6)
        np.testing.assert_allclose(env.feet_height, fx["feet_height"][t], rtol=0, atol=1e-5)
        np.testing.assert_allclose(np.stack([env.episode_sums[k] for k in env.reward_order]), fx["episode_sums"][t], rtol=0, atol=1e-5)
        if "levels" in fx:                               # terrain curriculum: rows and re-based origins after every reset
            np.testing.assert_array_equal(env.terrain_levels, fx["levels"][t], err_msg=f"levels step {t}")
            np.testing.assert_array_equal(env.env_origins, fx["origins"][t], err_msg=f"origins step {t}")
        if (t + 1) in full:
            np.testing.assert_allclose(obs, fx["full_obs"][full[t + 1]], rtol=0, atol=1e-4)
            np.testing.assert_allclose(priv, fx["full_priv"][full[t + 1]], rtol=0, atol=1e-4)
        if t >= 30:
            # free run for the first 30 steps; after that the recorded float64 physics state is loaded back every step, so
            # that fp32 round-off of the glue (torch there, numpy here) is not amplified by 64 chaotic robots over time
            st = env.state
            r = fx["root"][t].astype(st.q.dtype)
            st.root_pos, st.root_quat, st.root_linvel, st.root_angvel = r[:, 0:3].copy(), r[:, 3:7].copy(), r[:, 7:10].copy(), r[:, 10:13].copy()
            st.q, st.qd = fx["q"][t].astype(st.q.dtype).copy(), fx["qd"][t].astype(st.q.dtype).copy()
    if name == "env_rollout_b":
        assert fx["timeout"].sum() == 3 and fx["reset"].sum() >= 3          # the fixture does exercise time-outs


def test_terrain_fixture_places_robots_on_the_reference_tiles():
    """Fixture C (mesh_type='trimesh'): env origins come from the tile map by (level, type) with
    type = floor(i / (N / num_cols)) -- legged_robot.py:687-697 -- and robots stand on non-flat tiles."""
    fx = np.load(os.path.join(GOLD, "env_rollout_c.npz"))
    n = int(fx["meta"][0])
    types = np.floor(np.arange(n) / (n / fx["terrain_origins"].shape[1])).astype(int)
    assert np.array_equal(fx["terrain_types"], types)
    assert np.array_equal(fx["init_env_origins"], fx["terrain_origins"][fx["terrain_levels"], types])
    assert fx["init_env_origins"][:, 2].max() > 1.0                  # pyramid tops
    assert np.abs(fx["init_start_pos"][:, :2] - fx["init_env_origins"][:, :2]).max() <= 1.0
    assert fx["reset"].sum() >= 4 and fx["timeout"].sum() >= 2
    assert np.abs(fx["packs"][1:, 29:31]).sum() > 0                  # reset xy offsets were drawn
    assert float(fx["terrain_level_stat"]) == pytest.approx(fx["terrain_levels"].mean())


def test_curriculum_fixture_shows_every_branch():
    """Fixture D (terrain.curriculum=True): the reference's _update_terrain_curriculum (legged_robot.py:399-419) moved
    robots up, down and -- past the last row -- to a drawn row; the constructor's reset moved nobody."""
    fx = np.load(os.path.join(GOLD, "env_rollout_d.npz"))
    rows = fx["terrain_origins"].shape[0]
    assert np.array_equal(fx["init_env_origins"], fx["terrain_origins"][fx["init_terrain_levels"], fx["terrain_types"]])
    prev, moves = fx["init_terrain_levels"].copy(), []
    for t in range(fx["levels"].shape[0]):
        changed = np.nonzero(fx["levels"][t] != prev)[0]
        assert set(changed) <= set(np.nonzero(fx["reset"][t])[0])           # rows change only at a reset
        for e in np.nonzero(fx["reset"][t])[0]:
            moves.append((int(prev[e]), int(fx["levels"][t, e]), float(fx["packs"][t + 1][75, e])))
        prev = fx["levels"][t]
    assert any(b == a + 1 for a, b, _ in moves) and any(b == a - 1 for a, b, _ in moves)
    wraps = [(a, b, u) for a, b, u in moves if a == rows - 1 and b == 0]        # 2 -> 0 can only be the random row
    assert wraps and all(int(u * rows) == b for a, b, u in wraps)
    assert fx["packs"].shape[1] == 76


def test_fixture_exercises_events():
    a = np.load(os.path.join(GOLD, "env_rollout_a.npz"))
    b = np.load(os.path.join(GOLD, "env_rollout_b.npz"))
    assert a["reset"].sum() >= 3                      # contact terminations
    assert np.abs(b["packs"][:, 14:19]).sum() > 0     # a push happened (common_step_counter hits 400)
    assert np.abs(b["packs"][1:, 11:14]).sum() > 0    # a command resample at ep_len % 800 == 0
