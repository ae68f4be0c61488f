"""Zero-shot evaluation of an exported actor on the HIP simulator (the loop of the reference's play.py:133-143 without
the viewer, at any robot count): fixed command, `steps` env steps, per-robot survival and tracking statistics.

Used by tests/test_gpu_fidelity.py (the actors the reference ships were trained against PhysX; how they fare here is
the physics-fidelity evidence for SURVEY row a4) and by tools/actor_rollout.py for parameter studies.
"""
import numpy as np

from .. import capi
from ..algo.ppo import PPO, ActorCritic
from ..envs.configs import HectorCfg
from ..envs.hector_env import HectorFreeEnv


def load_actor_npz(path):
    """tests/golden/actors/<name>.npz -> {'actor.0.weight': ..., ...}"""
    d = np.load(path)
    return {"actor." + k: np.ascontiguousarray(d[k], np.float32) for k in d.files}


def load_actor_checkpoint(path):
    """model_<it>.pt of the runner (reference on_policy_runner.py:278-287) -> the actor.* tensors as float32 arrays"""
    import torch
    sd = torch.load(path, map_location="cpu", weights_only=False)["model_state_dict"]
    return {k: np.ascontiguousarray(v.numpy(), np.float32) for k, v in sd.items() if k.startswith("actor.")}


TILE_KINDS = ["flat", "obstacles", "rough", "slope up", "slope down", "stairs up", "stairs down"]


def roll_actor(actor_sd, num_envs=4096, steps=1000, command=(0.5, 0.0, 0.0, 0.0), mesh_type="plane", seed=11,
               cfg_edit=None, phys=None, warm=100, device="cuda:0", diagnostics=False, terrain_flags=0, by_tile=False):
    """Returns a dict of statistics.  A robot "falls" when its episode ends before the time limit (contact termination
    or blow-up guard); statistics of a robot stop at its first fall.  cfg_edit(cfg): optional config changes;
    phys: optional overrides of the contact-model constants (isaac_amd.envs.hector_env.PHYS); command=None keeps the env's
    own resampled commands; terrain_flags: hx_sim_set_terrain_options (1 = cliff cells keep their ramp, 2 = no sideways
    wall contact); by_tile (trimesh maps): adds res["tiles"] = rows (kind, difficulty tercile, robots, survival, all falls per
    robot per 10 s -- a fallen robot restarts on its tile's platform and keeps being counted)."""
    from ..envs import hector_env as he
    cfg = HectorCfg()
    cfg.env.num_envs = num_envs
    cfg.terrain.mesh_type = mesh_type
    cfg.seed = seed
    if cfg_edit is not None:
        cfg_edit(cfg)
    import torch
    torch.manual_seed(seed)
    np.random.seed(seed)
    saved = dict(he.PHYS)
    if phys:
        he.PHYS.update(phys)
    try:
        env = HectorFreeEnv(cfg, sim_device=device, headless=True)
    finally:
        he.PHYS.clear()
        he.PHYS.update(saved)
    if terrain_flags:
        capi.check(capi.lib().hx_sim_set_terrain_options(env._h, int(terrain_flags)), "hx_sim_set_terrain_options")
    ac = ActorCritic(env.num_obs, env.num_privileged_obs, env.num_actions, actor_hidden_dims=[512, 256, 128],
                     critic_hidden_dims=[768, 256, 128])
    alg = PPO(ac, device=device)
    alg.init_storage(num_envs, 1, [env.num_obs], [env.num_privileged_obs], [env.num_actions], obs_ld=env.obs_ld, priv_ld=env.priv_ld)
    sd = ac.state_dict()
    sd.update(actor_sd)
    ac.load_state_dict(sd)
    n = num_envs
    cmd = None if command is None else np.tile(np.asarray(command, np.float32), (n, 1))
    falls = np.zeros(n)
    alive = np.ones(n, bool)
    first_fall = np.full(n, steps, np.int64)
    vx_sum, vy_sum, wz_sum, z_sum, cnt, slip_sum, slip_cnt, sat_sum = (np.zeros(n) for _ in range(8))
    x0 = None
    obs = env.get_observations()
    for t in range(steps):
        act = ac.act_inference(obs)
        if cmd is not None:
            env.commands = cmd
        obs, _, _, dones, _ = env.step(act)
        d = dones.numpy().astype(bool)
        to = env.time_out_buf.numpy().astype(bool)
        fell = d & ~to
        falls += fell
        first_fall = np.where(alive & fell, t, first_fall)
        alive &= ~fell
        if t >= warm:
            lin, ang = env._base_velocities()
            root = env.get_state()[0]
            if x0 is None:
                x0 = root[:, :2].copy()
            m = alive
            vx_sum[m] += lin[m, 0]; vy_sum[m] += lin[m, 1]; wz_sum[m] += ang[m, 2]; z_sum[m] += root[m, 2]; cnt[m] += 1
            if diagnostics:
                bodies = env._buf(capi.BUF_BODY_STATE, (4, 13, n)).numpy()          # L_calf, L_toe, R_calf, R_toe
                cf = env.contact_forces
                tq = env.torques
                for k, b in ((0, 1), (1, 3)):
                    inc = cf[:, env.feet_indices[k], 2] > 5.0
                    sp = np.hypot(bodies[b, 7], bodies[b, 8])
                    slip_sum[m & inc] += sp[m & inc]; slip_cnt[m & inc] += 1
                sat_sum[m] += (np.abs(tq[m]) >= env.torque_limits[None, :] - 1e-3).mean(axis=1)
    ok = cnt > (steps - warm) // 2
    res = dict(num_envs=n, steps=steps, command=None if command is None else list(map(float, command)),
               falls_per_robot_10s=float(falls.sum() / n / (steps * env.dt) * 10.0),
               survival=float(alive.mean()),
               median_first_fall=float(np.median(first_fall)),
               mean_vx=float((vx_sum[ok] / cnt[ok]).mean()) if ok.any() else float("nan"),
               p10_vx=float(np.percentile(vx_sum[ok] / cnt[ok], 10)) if ok.any() else float("nan"),
               p90_vx=float(np.percentile(vx_sum[ok] / cnt[ok], 90)) if ok.any() else float("nan"),
               mean_vy=float((vy_sum[ok] / cnt[ok]).mean()) if ok.any() else float("nan"),
               mean_wz=float((wz_sum[ok] / cnt[ok]).mean()) if ok.any() else float("nan"),
               mean_height=float((z_sum[ok] / cnt[ok]).mean()) if ok.any() else float("nan"))
    if diagnostics:
        res["stance_foot_speed"] = float(slip_sum.sum() / max(1.0, slip_cnt.sum()))
        res["stance_fraction_per_foot"] = float(slip_cnt[ok].sum() / max(1.0, 2 * cnt[ok].sum()))
        res["torque_saturation_fraction"] = float(sat_sum[ok].sum() / max(1.0, cnt[ok].sum()))
        fr, ms = env.env_frictions, env.body_mass
        for nm, arr, edges in (("friction", fr, [0.1, 0.3, 0.5, 0.7, 1.01]), ("base_mass", ms, [6.0, 7.5, 9.0, 10.5, 12.3])):
            res["survival_by_" + nm] = [[edges[i], edges[i + 1], float(alive[(arr >= edges[i]) & (arr < edges[i + 1])].mean()) if ((arr >= edges[i]) & (arr < edges[i + 1])).any() else None]
                                        for i in range(len(edges) - 1)]
    if by_tile and getattr(env, "terrain", None) is not None and getattr(env, "_terrain_levels0", None) is not None:
        log = np.array(env.terrain.tile_log).reshape(cfg.terrain.num_rows, cfg.terrain.num_cols, 2)
        lv, ty = np.asarray(env.terrain_levels), np.asarray(env.terrain_types)
        kind, diff = log[lv, ty, 0].astype(int), log[lv, ty, 1]
        terc = np.minimum((diff * 3).astype(int), 2)
        rows = []
        for k in range(len(TILE_KINDS)):
            for t in (0, 1, 2, -1):
                m = (kind == k) & ((terc == t) if t >= 0 else True)
                if m.any():
                    rows.append(dict(kind=TILE_KINDS[k], tercile=("easy", "mid", "hard", "all")[t], robots=int(m.sum()), survival=float(alive[m].mean()),
                                     falls_per_robot_10s=float(falls[m].sum() / m.sum() / (steps * env.dt) * 10.0)))
        res["tiles"] = rows
    alg.close()
    env.close()
    return res
