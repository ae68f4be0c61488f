#!/usr/bin/env python3
"""Generate tests/golden/env_rollout_*.npz by running the REFERENCE env glue.

What runs: the reference's unmodified `HectorFreeEnv` (humanoid/envs/custom/hector_env.py +
humanoid/envs/base/legged_robot.py) imported from /root/reference over the stub isaacgym in
tests/refstub/, whose `gym.simulate` is oracle/physics.py.  Every random draw the reference makes
(torch.rand / torch.randn_like, reference hector_env.py:166-168,243, legged_robot.py:327-335,366,384,
hector_env.py:58-63) is recorded and laid out as the per-step "random pack" the product's parity
mode injects (include/hx_sim.h, HX_RP_* offsets).

Outputs per step: actions in; newest obs frame (41) and privileged frame (70); reward; reset and
time-out flags; post-step physics state (float64) and the tensors the glue read; selected full
615/1050 stacks.  Only data is stored -- no reference source text.

Run in this container only:  python tests/golden/make_env_fixtures.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
from tests.refstub import loader  # noqa: E402

# random-pack field offsets (floats per env); mirrored in include/hx_sim.h
RP = dict(delay=0, act_noise=1, cmd_a=11, push=14, reset_q=19, reset_xy=29, cmd_b=31, obs_noise=34, level=75)
RP_SIZE = 75          # + 1 row (RP["level"]) when the terrain curriculum is on


def generate(name, n_envs, n_steps, seed, action_std, ep_len_init=None, step_counter_init=0, add_noise=True,
             full_stack_steps=(0, 1, 14, 15, 16), terrain=None, reward_scales=None, cfg_overrides=None, task="hector"):
    """terrain=None: ground plane.  terrain=dict(mesh_type=, num_rows=, num_cols=, border_size=): the reference's
    own HumanoidTerrain (humanoid/utils/terrain.py) lays out the map, the env takes its origins from it
    (legged_robot.py:687-697) and every reset adds U[-1,1] to xy (:381-384)."""
    env_mod, cfg_mod, helpers = loader.load_env(task)
    from isaacgym import gymapi, torch_utils
    import isaacgym.torch_utils  # noqa: F401
    from oracle.env import HECTOR, HECTOR_FULL, HUMANOID
    T = {"hector": HECTOR, "hector_full": HECTOR_FULL, "humanoid_ppo": HUMANOID}[task]
    RP, RP_SIZE, ND, NOBS, NPRIV = T.rp, T.rp_size, T.ndof, T.nobs, T.npriv
    env_cls = {"hector": "HectorFreeEnv", "hector_full": "HectorFullFreeEnv", "humanoid_ppo": "XBotLFreeEnv"}[task]
    env_cls = getattr(env_mod, env_cls)

    cfg = getattr(cfg_mod, {"hector": "HectorCfg", "hector_full": "HectorFullCfg", "humanoid_ppo": "XBotLCfg"}[task])()
    cfg.terrain.mesh_type = "plane"
    if terrain is not None:
        for k, v in terrain.items():
            setattr(cfg.terrain, k, v)
    for k, v in (reward_scales or {}).items():      # e.g. the four reward terms HectorCfg zero-scales
        setattr(cfg.rewards.scales, k, v)
    for path, v in (cfg_overrides or {}).items():   # dotted attribute paths, e.g. "commands.heading_command"
        obj = cfg
        parts = path.split(".")
        for a in parts[:-1]:
            obj = getattr(obj, a)
        setattr(obj, parts[-1], v)
    cfg.env.num_envs = n_envs
    gymapi.Gym.trimesh_wall_height = float(cfg.terrain.slope_treshold) * float(cfg.terrain.horizontal_scale) if cfg.terrain.mesh_type == "trimesh" else 0.0
    cfg.noise.add_noise = add_noise
    cfg.seed = seed
    helpers.set_seed(seed)
    sim_params = gymapi.SimParams()
    sim_params.dt = cfg.sim.dt
    sim_params.use_gpu_pipeline = False

    log = []
    real_rand, real_randn_like, real_randint_like = torch.rand, torch.randn_like, torch.randint_like
    curriculum = bool(terrain is not None and terrain.get("curriculum", False))

    def randint_like(x, *a, **k):
        r = real_randint_like(x, *a, **k)
        log.append(("randint", r.clone(), int(a[0])))
        return r

    def rand(*a, **k):
        r = real_rand(*a, **k)
        log.append(("rand", r.clone()))
        return r

    def randn_like(x, *a, **k):
        r = real_randn_like(x, *a, **k)
        log.append(("randn", r.clone()))
        return r

    torch.rand, torch.randn_like, torch.randint_like = rand, randn_like, randint_like
    try:
        env = env_cls(cfg, sim_params, gymapi.SIM_PHYSX, "cpu", True)
        gym = env.gym
        gym.env = env
        N = n_envs
        # creation-time draws (start xy, friction buckets) are creation inputs, not step packs; the
        # constructor's reset_idx(all)+compute_observations made exactly the last five draws
        creation_draws = len(log) - (5 if add_noise else 4) - (1 if terrain is not None else 0)

        # wrap methods to leave markers with the env ids they act on
        orig_resample, orig_reset_dofs = env._resample_commands, env._reset_dofs

        def resample(ids):
            log.append(("mark_resample", ids.clone()))
            return orig_resample(ids)

        def reset_dofs(ids):
            log.append(("mark_reset", ids.clone()))
            return orig_reset_dofs(ids)

        env._resample_commands, env._reset_dofs = resample, reset_dofs

        def build_pack(entries, is_init):
            pack = np.zeros((RP_SIZE + (1 if curriculum else 0), N), np.float32)
            pending_level = None
            it = iter(entries)
            state = "start"
            resample_count = 0
            e = next(it, None)
            if not is_init:
                assert e[0] == "rand" and e[1].shape == (N, 1)
                pack[RP["delay"]] = e[1][:, 0].numpy()
                e = next(it)
                assert e[0] == "randn" and e[1].shape == (N, ND)
                pack[RP["act_noise"]:RP["act_noise"] + ND] = e[1].numpy().T
                e = next(it, None)
            while e is not None:
                if e[0] == "mark_resample":
                    ids = e[1].numpy()
                    field = RP["cmd_b"] if (state == "reset" or is_init) else RP["cmd_a"]
                    for k in range(3):
                        r = next(it)
                        assert r[0] == "rand" and r[1].shape == (len(ids), 1), (r[0], r[1].shape, len(ids))
                        pack[field + k, ids] = r[1][:, 0].numpy()
                    resample_count += 1
                elif e[0] == "randint":
                    # _update_terrain_curriculum (legged_robot.py:415): one draw per resetting env, made BEFORE _reset_dofs;
                    # stored as the uniform whose floor(u * max_level) is the drawn level
                    pending_level = (e[1].numpy().astype(np.float64) + 0.5) / e[2]
                elif e[0] == "mark_reset":
                    ids = e[1].numpy()
                    if pending_level is not None:
                        assert pending_level.shape == (len(ids),)
                        pack[RP["level"], ids] = pending_level.astype(np.float32)
                        pending_level = None
                    r = next(it)
                    assert r[0] == "rand" and r[1].shape == (len(ids), ND)
                    pack[RP["reset_q"]:RP["reset_q"] + ND, ids] = r[1].numpy().T
                    if terrain is not None:          # custom origins: xy offset of the reset pose
                        r = next(it)
                        assert r[0] == "rand" and r[1].shape == (len(ids), 2)
                        pack[RP["reset_xy"]:RP["reset_xy"] + 2, ids] = r[1].numpy().T
                    state = "reset"
                elif e[0] == "rand" and e[1].shape == (N, 2):
                    pack[RP["push"]:RP["push"] + 2] = e[1].numpy().T
                    r = next(it)
                    assert r[0] == "rand" and r[1].shape == (N, 3)
                    pack[RP["push"] + 2:RP["push"] + 5] = r[1].numpy().T
                elif e[0] == "randn" and e[1].shape == (N, NOBS):
                    pack[RP["obs_noise"]:RP["obs_noise"] + NOBS] = e[1].numpy().T
                else:
                    raise AssertionError(("unexpected draw", e[0], getattr(e[1], "shape", None)))
                e = next(it, None)
            return pack

        # the constructor already did reset_idx(all) + compute_observations (reference hector_env.py:50-51),
        # but before the markers were installed: re-derive the init pack from the raw log by position
        init_entries = log[creation_draws:]
        # order: rand(N,10) reset dofs ; 3x rand(N,1) resample ; randn(N,41)
        k0 = 2 if terrain is not None else 1
        ie = [("mark_reset", torch.arange(N))] + init_entries[:k0] + [("mark_resample", torch.arange(N))] + init_entries[k0:]
        packs = [build_pack(ie, True)]
        log.clear()

        out = {k: [] for k in ("actions", "obs41", "priv70", "rew", "reset", "timeout", "root", "q", "qd",
                               "contact", "bodies", "torques", "commands", "ep_len", "feet_air_time",
                               "feet_height", "episode_sums", "timeouts_visible")}
        if curriculum:
            out["levels"], out["origins"] = [], []
        full = {}
        init = dict(obs_full=env.obs_buf.numpy().copy(), priv_full=env.privileged_obs_buf.numpy().copy(),
                    root=gym.root_t.numpy().copy(), q=gym.state.q.copy(), commands=env.commands.numpy().copy(),
                    shape_friction=np.array(gym.shape_friction), base_mass=np.array(gym.base_mass),
                    start_pos=np.array(gym.start_pos),
                    env_origins=env.env_origins.numpy().copy(), env_frictions=env.env_frictions.numpy().copy(),
                    body_mass=env.body_mass.numpy().copy())
        if curriculum:
            init["terrain_levels"] = env.terrain_levels.numpy().copy()
        if ep_len_init is not None:
            env.episode_length_buf[:] = torch.as_tensor(ep_len_init, dtype=torch.long)
        env.common_step_counter = step_counter_init
        arng = np.random.default_rng(seed + 1000)
        reward_names = list(env.reward_names)
        for t in range(n_steps):
            a = (arng.standard_normal((N, ND)) * action_std).astype(np.float32)
            obs, priv, rew, reset, extras = env.step(torch.from_numpy(a.copy()))
            packs.append(build_pack(list(log), False))
            log.clear()
            out["actions"].append(a)
            out["obs41"].append(obs[:, -NOBS:].numpy().copy())         # newest frame (41 / 70 wide for hector, 65 / 94 for hector_full)
            out["priv70"].append(priv[:, -NPRIV:].numpy().copy())
            out["rew"].append(rew.numpy().copy())
            out["reset"].append(reset.numpy().astype(np.uint8))
            out["timeout"].append(env.time_out_buf.numpy().astype(np.uint8))
            tv = extras.get("time_outs")
            out["timeouts_visible"].append(tv.numpy().astype(np.uint8) if tv is not None else np.zeros(N, np.uint8))
            s = gym.state
            out["root"].append(np.concatenate([s.root_pos, s.root_quat, s.root_linvel, s.root_angvel], 1))
            out["q"].append(s.q.copy())
            out["qd"].append(s.qd.copy())
            out["contact"].append(env.contact_forces.numpy().copy())
            out["bodies"].append(env.rigid_state.numpy()[:, [T.knees[0], T.feet[0], T.knees[1], T.feet[1]]].copy())
            out["torques"].append(env.torques.numpy().copy())
            out["commands"].append(env.commands.numpy().copy())
            out["ep_len"].append(env.episode_length_buf.numpy().copy())
            out["feet_air_time"].append(env.feet_air_time.numpy().copy())
            out["feet_height"].append(env.feet_height.numpy().copy())
            out["episode_sums"].append(np.stack([env.episode_sums[k].numpy() for k in reward_names], 0))
            if curriculum:
                out["levels"].append(env.terrain_levels.numpy().copy())
                out["origins"].append(env.env_origins.numpy().copy())
            if (t + 1) in full_stack_steps or t == n_steps - 1:
                full[t + 1] = (obs.numpy().copy(), priv.numpy().copy())
        res = {k: np.stack(v) for k, v in out.items()}
        res["packs"] = np.stack(packs)
        for k, v in init.items():
            res["init_" + k] = v
        res["full_steps"] = np.array(sorted(full))
        res["full_obs"] = np.stack([full[k][0] for k in sorted(full)])
        res["full_priv"] = np.stack([full[k][1] for k in sorted(full)])
        res["task"] = np.array(task)
        res["reward_names"] = np.array(reward_names)
        res["cfg_override_names"] = np.array(sorted(cfg_overrides or {}), dtype="U64")
        res["cfg_override_values"] = np.array([json.dumps((cfg_overrides or {})[k]) for k in sorted(cfg_overrides or {})], dtype="U64")
        res["reward_override_names"] = np.array(sorted(reward_scales or {}), dtype="U32")
        res["reward_override_values"] = np.array([(reward_scales or {})[k] for k in sorted(reward_scales or {})], np.float64)
        res["reward_scales"] = np.array([env.reward_scales[k] for k in reward_names], np.float64)
        res["meta"] = np.array([n_envs, n_steps, seed, step_counter_init, int(add_noise)])
        res["ep_len_init"] = np.zeros(N, np.int64) if ep_len_init is None else np.asarray(ep_len_init, np.int64)
        res["noise_scale_vec"] = env.noise_scale_vec.numpy().copy()
        res["torque_limits"] = env.torque_limits.numpy().copy()
        res["p_gains"] = env.p_gains[0].numpy().copy()
        res["d_gains"] = env.d_gains[0].numpy().copy()
        res["default_dof_pos"] = env.default_dof_pos[0].numpy().copy()
        if terrain is not None:
            res["terrain_heights"] = np.asarray(env.terrain.heightsamples).copy()
            res["terrain_params"] = np.array([cfg.terrain.horizontal_scale, cfg.terrain.vertical_scale, cfg.terrain.border_size])
            res["terrain_wall_height"] = np.array(gymapi.Gym.trimesh_wall_height)
            res["terrain_levels"] = env.terrain_levels.numpy().copy()
            res["terrain_types"] = env.terrain_types.numpy().copy()
            res["terrain_origins"] = env.terrain_origins.numpy().copy()
            res["terrain_env_length"] = np.array(float(env.terrain.env_length))
            res["terrain_curriculum"] = np.array(int(curriculum))
            res["terrain_level_stat"] = np.array(float(env.extras["episode"]["terrain_level"])) if "episode" in env.extras and "terrain_level" in env.extras["episode"] else np.array(np.nan)
        path = os.path.join(HERE, name + ".npz")
        if getattr(gym.phys, "self_pairs", None):
            res["self_contact_count"] = np.int64(gym.phys.self_contact_count)
            print(name, "self-collision: %d active (robot, pair side, substep) triples" % gym.phys.self_contact_count)
        np.savez_compressed(path, **res)
        print(name, "steps", n_steps, "resets", int(res["reset"].sum()), "timeouts", int(res["timeout"].sum()),
              "torque checks", gym.torque_checks, "size %.0f KB" % (os.path.getsize(path) / 1024),
              "mean rew %.4f" % res["rew"].mean())
    finally:
        torch.rand, torch.randn_like, torch.randint_like = real_rand, real_randn_like, real_randint_like


if __name__ == "__main__":
    assert loader.available(), "needs /root/reference"
    only = set(sys.argv[1:])          # e.g. `make_env_fixtures.py env_rollout_d`; nothing = all
    want = lambda name: not only or name in only
    N = 8
    # A: ordinary rollout from the initial reset; falls (contact terminations) happen on their own
    if want("env_rollout_a"):
        generate("env_rollout_a", 64, 60, seed=5, action_std=2.5)      # 64 robots (SURVEY 8d config 1's env count)
    # B: exercises the calendar events: command resampling (ep_len % 800 == 0), time-outs (> 2400),
    #    the global push (counter % 400 == 0); observation noise off so stacks are exact
    if want("env_rollout_b"):
        generate("env_rollout_b", N, 40, seed=7, action_std=0.3,
                 ep_len_init=[795, 2396, 0, 799, 2399, 1599, 10, 2390], step_counter_init=390, add_noise=False)
    # C: rough terrain (the reference's default mesh_type): tile map from the reference's HumanoidTerrain,
    #    env origins on the tiles, reset xy offsets, contact against slopes / blocks / stairs
    if want("env_rollout_c"):
        generate("env_rollout_c", 64, 50, seed=5, action_std=0.6,
                 ep_len_init=[0, 2350, 0, 0, 2380, 0, 0, 0] * 8,
                 terrain=dict(mesh_type="trimesh", num_rows=2, num_cols=4, border_size=3.0))
    # E: the four reward terms that HectorCfg zero-scales (joint_pos, low_speed, track_vel_hard, vel_mismatch_exp), switched
    #    on with the scales the sibling config hector_w_arm_config.py uses (:178-182) and humanoid_config's joint_pos 1.6:
    #    the product's kernel implements them, this pins them
    if want("env_rollout_e"):
        generate("env_rollout_e", N, 90, seed=23, action_std=0.5,
                 ep_len_init=[0, 37, 2385, 5, 797, 63, 31, 2395],
                 reward_scales=dict(joint_pos=1.6, low_speed=0.2, track_vel_hard=0.5, vel_mismatch_exp=0.5))
    # F: config branches HectorCfg never takes: yaw-rate commands instead of heading commands (legged_robot.py:329-332,
    #    :310-313 skipped), rewards not clipped at zero (:226-227), no pushes (:318), the sibling config's command ranges,
    #    a non-zero action delay and another action-noise level (hector_env.py:166-168)
    if want("env_rollout_f"):
        generate("env_rollout_f", N, 60, seed=31, action_std=0.4,
                 ep_len_init=[795, 2396, 0, 799, 2399, 1599, 10, 2390], step_counter_init=390,
                 cfg_overrides={"commands.heading_command": False, "rewards.only_positive_rewards": False,
                                "domain_rand.push_robots": False, "commands.ranges.lin_vel_x": [-0.6, 0.8],
                                "commands.ranges.ang_vel_yaw": [-0.5, 0.5],
                                "domain_rand.action_delay": 0.5, "domain_rand.action_noise": 0.05})
    # G: the sibling task hector_full (18 DoF: legs + arms, reference hector_w_arm_env.py / hector_w_arm_config.py) through
    #    the same stub -- pins the oracle's restatement of that task's glue (SURVEY 8f-4 groundwork; no kernel yet)
    if want("env_rollout_g"):
        generate("env_rollout_g", N, 80, seed=37, action_std=0.5, task="hector_full",
                 ep_len_init=[795, 2396, 0, 799, 2399, 1599, 10, 2390], step_counter_init=390)
    # H: the sibling task humanoid_ppo (XBot-L, 12 DoF, joints about the z axes of rotated frames; reference humanoid_env.py /
    #    humanoid_config.py) through the same stub: 47-wide frames x 15, the 73-wide privileged frame x 3, the joint_pos reward that
    #    follows the gait reference (default pose = 0, so the term is sensitive to the reference's phase), action delay 0.5
    if want("env_rollout_h"):
        generate("env_rollout_h", N, 80, seed=41, action_std=2.0, task="humanoid_ppo",
                 ep_len_init=[795, 2396, 0, 799, 2399, 1599, 10, 2390], step_counter_init=390)
    # D: terrain curriculum (legged_robot.py:399-419) on a 3 x 2 map of 1.6 m tiles: the reset xy offset alone carries
    #    about half of the robots past env_length / 2 = 0.8 m (move up; past the last row -> a random row), the others
    #    fall short of half their commanded distance (move down) or, with a zero command, stay.  Seed 21 shows every
    #    branch: 0->1 and 1->2 (up), 1->0 and 2->1 (down), 2->0 twice (past the last row -> random row)
    if want("env_rollout_d"):
        generate("env_rollout_d", N, 150, seed=21, action_std=0.8,
                 ep_len_init=[2290, 2300, 2310, 2320, 0, 2340, 0, 2360],
                 terrain=dict(mesh_type="trimesh", curriculum=True, num_rows=3, num_cols=2, border_size=2.0,
                              terrain_length=1.6, terrain_width=1.6, max_init_terrain_level=2))
