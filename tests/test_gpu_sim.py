"""GPU: the HIP env step against the golden trajectories produced by the reference's own env glue over
the float64 oracle physics (tests/golden/env_rollout_*.npz), through the C ABI.

The kernel integrates in fp32 with a different algorithm (ABA) than the oracle (CRBA, float64), so:
  * teacher-forced single steps (physics state reloaded from the fixture before every step) must match
    tightly: observation error median < 1e-4, 90th percentile < 2e-3, max < 2e-2; rewards 2e-4;
    reset / time-out flags and episode lengths exact;
  * a free run must stay within 2e-2 on observations for the first 30 steps (contact-rich chaos amplifies
    fp32 round-off afterwards; the drift is reported).
"""
import os

import numpy as np
import pytest

from tests.step_errors import check_step_errors

from isaac_amd import capi
from isaac_amd.envs.configs import HectorCfg, HectorFullCfg
from isaac_amd.envs.hector_env import HectorFreeEnv, HectorFullFreeEnv

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def make_env(fx):
    n, steps, seed, sc0, noise = (int(x) for x in fx["meta"])
    task = str(fx["task"]) if "task" in fx else "hector"             # fixture G: hector_full, fixture H: humanoid_ppo (XBot-L)
    from isaac_amd.envs.configs import XBotLCfg
    from isaac_amd.envs.hector_env import XBotLFreeEnv
    cfg_cls, env_cls = {"hector": (HectorCfg, HectorFreeEnv), "hector_full": (HectorFullCfg, HectorFullFreeEnv),
                        "humanoid_ppo": (XBotLCfg, XBotLFreeEnv)}[task]
    cfg = cfg_cls()
    cfg.env.num_envs = n
    cfg.noise.add_noise = bool(noise)
    cfg.seed = seed
    cfg.terrain.mesh_type = "plane"
    if "cfg_override_names" in fx:        # fixture F: config branches HectorCfg never takes
        import json
        for path, v in zip(fx["cfg_override_names"], fx["cfg_override_values"]):
            obj, parts = cfg, str(path).split(".")
            for a in parts[:-1]:
                obj = getattr(obj, a)
            setattr(obj, parts[-1], json.loads(str(v)))
    if "reward_override_names" in fx:     # fixture E: the four terms HectorCfg zero-scales, switched on
        for k, v in zip(fx["reward_override_names"], fx["reward_override_values"]):
            setattr(cfg.rewards.scales, str(k), float(v))
    creation = dict(friction=fx["init_shape_friction"], mass=fx["init_base_mass"], origins=fx["init_env_origins"],
                    start=fx["init_start_pos"])
    if "terrain_heights" in fx:           # fixture C: the tile map the reference's HumanoidTerrain laid out
        hs, vs, border = (float(x) for x in fx["terrain_params"])
        cfg.terrain.mesh_type = "trimesh"
        creation["terrain"] = dict(heights=fx["terrain_heights"], horizontal_scale=hs, vertical_scale=vs, border_size=border)
        creation["terrain_levels"], creation["terrain_types"] = fx["terrain_levels"], fx["terrain_types"]
        if "terrain_curriculum" in fx and int(fx["terrain_curriculum"]):      # fixture D: legged_robot.py:399-419
            cfg.terrain.curriculum = True
            cfg.terrain.num_rows, cfg.terrain.num_cols = (int(x) for x in fx["terrain_origins"].shape[:2])
            cfg.terrain.terrain_length = cfg.terrain.terrain_width = float(fx["terrain_env_length"])
            creation["terrain_levels"], creation["terrain_origins"] = fx["init_terrain_levels"], fx["terrain_origins"]
    env = env_cls(cfg, sim_device="cuda:0", creation=creation, init_pack=fx["packs"][0])
    return env, n, steps, sc0


@pytest.mark.parametrize("name", ["env_rollout_a", "env_rollout_b", "env_rollout_c", "env_rollout_d", "env_rollout_g", "env_rollout_h"])
def test_constructor_reset_and_first_observation(hxlib, name):
    fx = np.load(os.path.join(GOLD, name + ".npz"))
    env, n, steps, sc0 = make_env(fx)
    np.testing.assert_allclose(env.obs_buf.numpy(), fx["init_obs_full"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(env.privileged_obs_buf.numpy(), fx["init_priv_full"], rtol=0, atol=2e-5)
    root, q, qd = env.get_state()
    np.testing.assert_allclose(q, fx["init_q"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(env.commands, fx["init_commands"], rtol=0, atol=1e-6)
    env.close()


@pytest.mark.parametrize("name", ["env_rollout_a", "env_rollout_b", "env_rollout_c", "env_rollout_d", "env_rollout_e", "env_rollout_f",
                                  "env_rollout_g", "env_rollout_h"])
def test_teacher_forced_steps(hxlib, name):
    fx = np.load(os.path.join(GOLD, name + ".npz"))
    env, n, steps, sc0 = make_env(fx)
    env.episode_length_buf = fx["ep_len_init"].astype(np.int32)
    env.set_step_counter(sc0)
    errs = dict(obs=[], priv=[], rew=[], tau=[], contact=[])      # per (step, robot)
    for t in range(steps):
        if t > 0:
            # reload the oracle's post-step physics state (after resets/pushes) so errors do not accumulate
            env.set_state(fx["root"][t - 1].astype(np.float32), fx["q"][t - 1].astype(np.float32), fx["qd"][t - 1].astype(np.float32))
        obs, priv, rew, reset, extras = env.step(fx["actions"][t], pack=fx["packs"][t + 1])
        o, p = obs.numpy(), priv.numpy()
        errs["obs"].append(np.abs(o[:, -env.obs_frame:] - fx["obs41"][t]).max(axis=1))
        errs["priv"].append(np.abs(p[:, -env.priv_frame:] - fx["priv70"][t]).max(axis=1))
        errs["rew"].append(np.abs(rew.numpy() - fx["rew"][t]))
        errs["tau"].append(np.abs(env.torques - fx["torques"][t]).max(axis=1))
        errs["contact"].append(np.abs(env.contact_forces - fx["contact"][t]).reshape(n, -1).max(axis=1))
        assert np.array_equal(reset.numpy(), fx["reset"][t]), f"reset flags differ at step {t}"
        assert np.array_equal(env.time_out_buf.numpy(), fx["timeout"][t]), f"time-out flags differ at step {t}"
        assert np.array_equal(extras["time_outs"].numpy(), fx["timeouts_visible"][t]), f"extras time_outs differ at step {t}"
        np.testing.assert_array_equal(env.episode_length_buf.numpy(), fx["ep_len"][t])
        if "levels" in fx:       # terrain curriculum: the row every robot is on after this step's resets
            np.testing.assert_array_equal(env.terrain_levels, fx["levels"][t], err_msg=f"terrain levels differ at step {t}")
    # observations, rewards, torques AND contact forces: the tight tier for every (step, robot) pair but a counted handful per
    # fixture, the loose one for all (tests/step_errors.py)
    check_step_errors(name + " HIP kernel", errs)
    env.close()


def test_free_run_drift(hxlib):
    fx = np.load(os.path.join(GOLD, "env_rollout_a.npz"))
    env, n, steps, sc0 = make_env(fx)
    per_robot = np.zeros(n)
    for t in range(20):
        obs, priv, rew, reset, extras = env.step(fx["actions"][t], pack=fx["packs"][t + 1])
        per_robot = np.maximum(per_robot, np.abs(obs.numpy()[:, -41:] - fx["obs41"][t]).max(axis=1))
    # 64 robots under unit-variance random actions are chaotic (tests/test_host_build.py has the CPU twin of this test):
    # the typical robot stays at round-off over 20 free-running steps, a few separate at a contact threshold
    print("free run, 20 steps: median %.2e, 90 %% %.2e, worst %.2e" % (np.median(per_robot), np.quantile(per_robot, 0.9), per_robot.max()))
    assert np.median(per_robot) < 2e-3 and np.quantile(per_robot, 0.9) < 3e-2
    env.close()


def test_full_stacks_and_history_zeroing(hxlib):
    """615/1050 stacks at recorded steps, including steps right after a reset (history rows zeroed, hector_env.py:256-261)."""
    fx = np.load(os.path.join(GOLD, "env_rollout_b.npz"))
    env, n, steps, sc0 = make_env(fx)
    env.episode_length_buf = fx["ep_len_init"].astype(np.int32)
    env.set_step_counter(sc0)
    full = {int(s): i for i, s in enumerate(fx["full_steps"])}
    for t in range(steps):
        if t > 0:
            env.set_state(fx["root"][t - 1].astype(np.float32), fx["q"][t - 1].astype(np.float32), fx["qd"][t - 1].astype(np.float32))
        obs, priv, *_ = env.step(fx["actions"][t], pack=fx["packs"][t + 1])
        if (t + 1) in full:
            i = full[t + 1]
            np.testing.assert_allclose(obs.numpy(), fx["full_obs"][i], rtol=0, atol=2e-3)
            np.testing.assert_allclose(priv.numpy(), fx["full_priv"][i], rtol=0, atol=5e-3)
    env.close()


def test_pipelined_shards_equal_single_env(hxlib):
    """PipelinedHectorEnv (2 shards, own streams) simulates exactly the robots of the unsharded env: same creation
    draws, same Philox streams (keyed by global env id) -> bit-identical observations, rewards and resets."""
    from isaac_amd.envs.hector_env import PipelinedHectorEnv
    from isaac_amd.utils.helpers import set_seed
    n = 64
    rng = np.random.default_rng(0)
    acts = (0.5 * rng.standard_normal((6, n, 10))).astype(np.float32)
    outs = []
    for sharded in (False, True):
        cfg = HectorCfg()
        cfg.env.num_envs = n
        cfg.seed = set_seed(11)
        cfg.terrain.mesh_type = "plane"
        env = PipelinedHectorEnv(cfg, num_shards=2) if sharded else HectorFreeEnv(cfg)
        rec = []
        for t in range(6):
            if sharded:
                res = env.step(acts[t])
                rec.append((np.concatenate([r[0].numpy() for r in res]), np.concatenate([r[2].numpy() for r in res]),
                            np.concatenate([r[3].numpy() for r in res])))
            else:
                o, p, r, d, _ = env.step(acts[t])
                rec.append((o.numpy(), r.numpy(), d.numpy()))
        outs.append(rec)
        env.close()
    for (o0, r0, d0), (o1, r1, d1) in zip(*outs):
        np.testing.assert_array_equal(o0, o1)
        np.testing.assert_array_equal(r0, r1)
        np.testing.assert_array_equal(d0, d1)


# ---------------------------------------------------------------------------------------------- rough terrain
def _rough_setup(n, seed):
    """n robots scattered over a small HumanoidTerrain map (all tile kinds), HIP env + oracle env on the same data."""
    from isaac_amd.envs.terrain import HumanoidTerrain
    from oracle.env import HectorEnvOracle
    from oracle.terrain import HeightField

    class T(HectorCfg.terrain):
        mesh_type, num_rows, num_cols, curriculum, border_size = "trimesh", 3, 10, True, 2.0
    np.random.seed(seed)
    ter = HumanoidTerrain(T, n)                 # curriculum layout: every tile kind, three difficulties
    rng = np.random.default_rng(seed)
    hf = HeightField(ter.heightsamples, 0.1, 0.005, 2.0, wall_height=0.075)        # mesh_type 'trimesh': slope_treshold 0.75 x 0.1 m
    origins = np.zeros((n, 3), np.float32)
    origins[:, 0] = rng.uniform(1.0, 23.0, n)
    origins[:, 1] = rng.uniform(1.0, 79.0, n)
    origins[:, 2] = hf.query(origins[:, 0], origins[:, 1])[0] - 0.05        # start with the feet slightly in the ground
    fr, ms = rng.uniform(0.3, 1.0, n).astype(np.float32), (8.15528 + rng.uniform(-2, 4, n)).astype(np.float32)
    pack = lambda: np.concatenate([rng.uniform(size=(34, n)), rng.standard_normal((41, n))]).astype(np.float32)
    p0 = pack()
    p0[29:31] = 0.5                             # reset xy offset 0: stand exactly on the chosen spot
    cfg = HectorCfg()
    cfg.env.num_envs = n
    cfg.terrain.mesh_type = "trimesh"
    grid = dict(heights=ter.heightsamples, horizontal_scale=0.1, vertical_scale=0.005, border_size=2.0)
    env = HectorFreeEnv(cfg, sim_device="cuda:0", init_pack=p0,
                        creation=dict(friction=fr, mass=ms, origins=origins, start=origins.copy(), terrain=grid))
    orc = HectorEnvOracle(n, fr, ms, origins, p0, start_xy=origins.copy(), terrain=hf, custom_origins=True)
    return env, orc, rng, pack, hf


def test_terrain_contact_matches_oracle(hxlib):
    """Robots standing / stepping on slopes, blocks, rough ground and stairs: every env step teacher-forced from the
    oracle's state must agree with it (same criteria as the plane fixtures), and feet must actually load the terrain."""
    n = 64
    env, orc, rng, pack, hf = _rough_setup(n, 4)
    np.testing.assert_allclose(env.obs_buf.numpy(), orc.obs_buf, atol=1e-5)
    errs, loaded, tilted = dict(obs=[], priv=[], rew=[], tau=[], contact=[]), 0, 0
    for t in range(25):
        if t > 0:
            s = orc.state
            root = np.concatenate([s.root_pos, s.root_quat, s.root_linvel, s.root_angvel], 1).astype(np.float32)
            env.set_state(root, s.q.astype(np.float32), s.qd.astype(np.float32))
        a, pk = (0.3 * rng.standard_normal((n, 10))).astype(np.float32), pack()
        obs, priv, rew, done, _ = env.step(a, pack=pk)
        o2, p2, r2, d2 = orc.step(a, pk)
        alive = ~(d2 | done.numpy().astype(bool))
        assert np.array_equal(done.numpy().astype(bool), d2), f"reset flags differ at step {t}"
        errs["obs"].append(np.abs(obs.numpy()[alive][:, -41:] - o2[alive][:, -41:]).max(axis=1))
        errs["priv"].append(np.abs(priv.numpy()[alive][:, -70:] - p2[alive][:, -70:]).max(axis=1))
        errs["rew"].append(np.abs(rew.numpy() - r2))
        errs["tau"].append(np.abs(env.torques - orc.torques)[alive].max(axis=1))
        cf = orc.phys.contact_force
        errs["contact"].append(np.abs(env.contact_forces - cf)[alive].reshape(int(alive.sum()), -1).max(axis=1))
        loaded += int((cf[:, [5, 10], 2] > 20.0).sum())
        tilted += int((np.abs(cf[:, [5, 10], :2]).max(-1) > 0.3 * np.abs(cf[:, [5, 10], 2]) + 1.0).sum())
    # 64 robots that start 5 cm inside the ground (violent first steps) on every tile kind, walls included
    check_step_errors("terrain contact, HIP kernel vs oracle", errs)
    print("loaded feet %d, on inclines %d" % (loaded, tilted))
    assert loaded > 500 and tilted > 20
    env.close()


def test_loaded_feet_rest_on_the_surface(hxlib):
    """Property check without the oracle's dynamics: while robots stand and sway on the tiles (zero actions, 0.4 s),
    every foot that carries load sits ON the height field under it -- toe origin 4 cm above its sole, so the gap to
    the surface is a few centimetres (more only where a sole bridges a stair edge) and never clearly negative."""
    n = 256
    env, orc, rng, pack, hf = _rough_setup(n, 9)
    zero = np.zeros((n, 10), np.float32)
    ever_reset = np.zeros(n, bool)
    gaps = []
    for t in range(40):
        _, _, _, done, _ = env.step(zero)
        ever_reset |= done.numpy().astype(bool)
        if t < 10:
            continue                                                    # let the initial 5 cm penetration resolve
        bodies = env._buf(capi.BUF_BODY_STATE, (4, 13, n)).numpy()          # L_calf, L_toe, R_calf, R_toe
        cf = env.contact_forces
        for k, body in ((1, 5), (3, 10)):
            sel = (~ever_reset) & (cf[:, body, 2] > 20.0)
            gaps.append(bodies[k, 2, sel] - hf.query(bodies[k, 0, sel], bodies[k, 1, sel])[0])
    gaps = np.concatenate(gaps)
    print("loaded-foot gap to the surface: n=%d min %.3f median %.3f p95 %.3f max %.3f"
          % (len(gaps), gaps.min(), np.median(gaps), np.quantile(gaps, 0.95), gaps.max()))
    assert len(gaps) > 2000
    assert gaps.min() > -0.03 and 0.01 < np.median(gaps) < 0.06 and np.quantile(gaps, 0.95) < 0.10
    env.close()


def test_blow_up_guard(hxlib):
    """A robot whose physics state is non-finite (or absurd) is contained: that robot's episode ends this step, every
    observation / reward stays finite, and the other robots are bit-identical to a run without the poisoned neighbours."""
    fx = np.load(os.path.join(GOLD, "env_rollout_a.npz"))
    outs = []
    for poison in (False, True):
        env, n, steps, sc0 = make_env(fx)
        root, q, qd = env.get_state()
        if poison:
            root[3, 0] = np.nan
            root[5, 7] = 3.0e9
            qd[6, 2] = np.inf
            env.set_state(root, q, qd)
        rec = []
        for t in range(3):
            o, p, r, d, _ = env.step(fx["actions"][t], pack=fx["packs"][t + 1])
            rec.append((o.numpy().copy(), p.numpy().copy(), r.numpy().copy(), d.numpy().copy()))
        outs.append(rec)
        env.close()
    clean, bad = outs
    assert bad[0][3][[3, 5, 6]].all() and not clean[0][3][[3, 5, 6]].any()          # the three robots were reset at once
    others = [i for i in range(8) if i not in (3, 5, 6)]
    for (o0, p0, r0, d0), (o1, p1, r1, d1) in zip(clean, bad):
        assert np.isfinite(o1).all() and np.isfinite(p1).all() and np.isfinite(r1).all()
        assert np.array_equal(o0[others], o1[others]) and np.array_equal(r0[others], r1[others]) and np.array_equal(d0[others], d1[others])


def test_curriculum_with_device_rng(hxlib):
    """terrain.curriculum=True through the ordinary constructor (curriculum tile layout, Philox draws): rows stay inside
    the table, change only when the robot resets, and every reset pose sits within 1 m (the U[-1,1] xy offset) of the
    platform origin of the robot's current (row, column) tile -- legged_robot.py:381-384, :399-419."""
    from isaac_amd.utils.helpers import set_seed
    n = 256
    cfg = HectorCfg()
    cfg.env.num_envs = n
    cfg.seed = set_seed(3)
    cfg.terrain.mesh_type = "trimesh"
    cfg.terrain.curriculum = True
    cfg.terrain.num_rows, cfg.terrain.num_cols, cfg.terrain.border_size = 4, 4, 5.0
    cfg.terrain.terrain_length = cfg.terrain.terrain_width = 4.0
    cfg.terrain.max_init_terrain_level = 1
    env = HectorFreeEnv(cfg, sim_device="cuda:0")
    lv0 = env.terrain_levels.copy()
    assert lv0.min() >= 0 and lv0.max() <= 1                      # max_init_terrain_level
    rng = np.random.default_rng(0)
    prev, moved = lv0, 0
    for t in range(120):
        obs, priv, rew, reset, extras = env.step((1.5 * rng.standard_normal((n, 10))).astype(np.float32))
        lv, r = env.terrain_levels, reset.numpy().astype(bool)
        assert lv.min() >= 0 and lv.max() < 4
        assert not np.any((lv != prev) & ~r)
        moved += int(np.sum(lv != prev))
        if r.any():
            root, _, _ = env.get_state()
            og = env.terrain_origins[lv, np.asarray(env.terrain_types)]
            assert np.abs(root[r, :2] - og[r, :2]).max() <= 1.0 + 1e-5
            np.testing.assert_allclose(root[r, 2], og[r, 2] + 0.55, rtol=0, atol=1e-5)
        prev = lv
    assert moved > 0
    info, cnt = env.episode_stats()
    assert info["terrain_level"] == pytest.approx(float(np.mean(env.terrain_levels)))
    env.close()


def test_zero_copy_step_writes_the_callers_rows_and_refuses_misaligned_ones(hxlib):
    """hx_sim_step_ex (include/hx_sim.h): the stacking launch writes observation t+1 and reward / done / time-out of step t straight
    into the caller's buffers with 16-byte stores -- the same values hx_sim_step leaves in the simulator's own buffers -- and a
    destination that is not 16-byte aligned is refused with a message instead of being written to."""
    from isaac_amd import capi
    from isaac_amd.envs.configs import HectorCfg
    from isaac_amd.envs.hector_env import HectorFreeEnv
    from isaac_amd.utils.helpers import set_seed
    N = 40                                   # five waves; not a multiple of anything else
    outs = []
    for zero_copy in (False, True):
        cfg = HectorCfg(); cfg.env.num_envs = N; cfg.seed = set_seed(3)
        env = HectorFreeEnv(cfg)
        act = capi.DeviceBuffer.from_host((0.2 * np.random.default_rng(1).standard_normal((N, 10))).astype(np.float32))
        L = env._L
        if zero_copy:
            ods = [capi.DeviceBuffer(N * env.obs_ld * 4 + 64) for _ in range(2)]
            pds = [capi.DeviceBuffer(N * env.priv_ld * 4 + 64) for _ in range(2)]
            rw, dn, to = capi.DeviceBuffer(N * 4), capi.DeviceBuffer(N), capi.DeviceBuffer(N)
            rc = L.hx_sim_step_ex(env._h, act.ptr, None, ods[0].ptr + 4, pds[0].ptr, rw.ptr, dn.ptr, to.ptr)
            assert rc != 0 and b"16-byte aligned" in L.hx_last_error()
            for t in range(3):                   # the rollout storage's pattern: slot t + 1 is written from slot t
                capi.check(L.hx_sim_step_ex(env._h, act.ptr, None, ods[t & 1].ptr, pds[t & 1].ptr, rw.ptr, dn.ptr, to.ptr), "step_ex")
            rc = L.hx_sim_step_ex(env._h, act.ptr, None, ods[0].ptr, pds[0].ptr, rw.ptr, dn.ptr, to.ptr)      # in place: the rows of the step before
            assert rc != 0 and b"buffer of their own" in L.hx_last_error()
            od, pd = ods[0], pds[0]
            env.sync()
            outs.append((od.download(np.float32, (N, env.obs_ld)), pd.download(np.float32, (N, env.priv_ld)), rw.download(np.float32, (N,)), dn.download(np.uint8, (N,))))
        else:
            for _ in range(3):
                capi.check(L.hx_sim_step(env._h, act.ptr, None), "step")
            env.sync()
            outs.append((env._buf(capi.BUF_OBS, (N, env.obs_ld)).numpy().copy(), env._buf(capi.BUF_PRIV, (N, env.priv_ld)).numpy().copy(),
                         env._buf(capi.BUF_REW, (N,)).numpy().copy(), env._buf(capi.BUF_RESET, (N,), np.uint8).numpy().copy()))
        env.close()
    for a, b, name in zip(outs[0], outs[1], ("obs rows", "privileged rows", "rewards", "dones")):
        np.testing.assert_array_equal(a, b, err_msg=name)
    assert np.abs(outs[0][0]).sum() > 0
