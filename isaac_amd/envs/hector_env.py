"""HectorFreeEnv: the reference's VecEnv for task `hector`, served by the HIP simulator (libhx.so).

Constructor signature and attributes follow the reference (humanoid/envs/custom/hector_env.py:46-51,
humanoid/envs/base/legged_robot.py:58-82, humanoid/envs/base/base_task.py:43-125) so that
`task_class(cfg=..., sim_params=..., physics_engine=..., sim_device=..., headless=...)`
(humanoid/utils/task_registry.py:97-101) keeps working.  All per-step arithmetic happens in
isaac_amd/csrc/hx_sim.hip; this file only
  * derives the flat C config from the nested config classes (what _parse_cfg / _init_buffers /
    _prepare_reward_function / _get_noise_scale_vec do: legged_robot.py:710-720,433-540, hector_env.py:135-155),
  * draws the creation-time randomisation on the host in the reference's call order
    (legged_robot.py:650-664: start xy per env, friction buckets at env 0, payload per env),
  * hands out DeviceArray views of the library's buffers.
"""
import math

import numpy as np

from .. import capi
from ..devarray import DeviceArray, device_pointer
from ..cfgtools import class_to_dict  # noqa: F401  (kept importable from here)
from .vec_env import VecEnv

DOF_NAMES = ["L_hip_joint", "L_hip_roll_joint", "L_thigh_joint", "L_calf_joint", "L_toe_joint",
             "R_hip_joint", "R_hip_roll_joint", "R_thigh_joint", "R_calf_joint", "R_toe_joint"]
BODY_NAMES = ["base", "L_hip", "L_hip2", "L_thigh", "L_calf", "L_toe", "R_hip", "R_hip2", "R_thigh", "R_calf", "R_toe"]
URDF_EFFORT = [33.5, 33.5, 33.5, 67.0, 33.5] * 2       # robot.urdf <limit effort=...>
BASE_MASS = 8.15528                                    # collapsed base link (tools/compile_urdf.py)

# physics model constants (DESIGN.md "Physics model"); mirrored in oracle/physics.py
PHYS = dict(contact_kn=4.0e4, contact_dn=4.0e2, friction_veps=2.0e-2, limit_k=2.0e3, limit_d=2.0e1)


def creation_randomisation(cfg, num_envs, env_origins, base_mass=BASE_MASS):
    """Per-env friction, base mass and start pose, drawn with the same generators in the same order as the
    reference's _create_envs loop (legged_robot.py:650-664 with callbacks :244-301), so equal seeds give
    equal robots.  torch's CPU generator is used when torch is importable (plumbing only)."""
    fr = cfg.domain_rand
    start = np.array(env_origins, np.float32).copy()
    friction = np.ones(num_envs, np.float32)
    mass = np.full(num_envs, base_mass, np.float32)
    try:
        import torch
        rand = lambda *s: torch.rand(*s).numpy()
        randint = lambda hi, n: torch.randint(0, hi, (n, 1)).numpy()[:, 0]
    except ImportError:                                   # pragma: no cover
        rand = lambda *s: np.random.random_sample(s).astype(np.float32)
        randint = lambda hi, n: np.random.randint(0, hi, n)
    coeffs = None
    for i in range(num_envs):
        start[i, :2] += (2.0 * rand(2, 1)[:, 0] - 1.0).astype(np.float32)
        if getattr(fr, "randomize_friction", False):
            if i == 0:
                buckets = randint(256, num_envs)
                lo, hi = fr.friction_range
                fb = ((hi - lo) * rand(256, 1) + lo)[:, 0].astype(np.float32)
                coeffs = fb[buckets]
            friction[i] = coeffs[i]
        if getattr(fr, "randomize_base_mass", False):
            lo, hi = fr.added_mass_range
            mass[i] = np.float32(base_mass + np.random.uniform(lo, hi))
    return friction, mass, start


class HectorFreeEnv(VecEnv):
    # what a sibling task of the family overrides (HectorFullFreeEnv, XBotLFreeEnv below)
    DOF_NAMES, BODY_NAMES, URDF_EFFORT, BASE_MASS = DOF_NAMES, BODY_NAMES, URDF_EFFORT, BASE_MASS
    PRIV_BASE, PRIV_STACK = 40, 15          # privileged frame = PRIV_BASE + 3 * num_dof values, PRIV_STACK frames per row
    # side-local body indices the kernel's model descriptor names (asset.knee_name / foot_name; checked against the config)
    KNEE_LOCAL, FOOT_LOCAL = 4, 5

    def __init__(self, cfg, sim_params=None, physics_engine=None, sim_device="cuda:0", headless=True, stream=None,
                 creation=None, init_pack=None, env_range=None):
        """env_range=(lo, hi): this object simulates envs lo..hi-1 of the logical batch of cfg.env.num_envs robots
        (used by PipelinedHectorEnv); creation-time draws are made for the whole batch and sliced."""
        self.cfg = cfg
        self.sim_params = sim_params
        self.headless = headless
        self.sim_device = sim_device
        self.device = sim_device
        L = capi.lib()
        if str(sim_device).startswith("cuda") and ":" in str(sim_device):
            capi.check(L.hx_set_device(int(str(sim_device).split(":")[1])), "hx_set_device")
        c, friction, mass, start, terrain_grid, rough = self._derive(cfg, sim_params, creation, env_range)
        self._create(L, c, friction, mass, start, terrain_grid, rough, stream, init_pack)

    def _derive(self, cfg, sim_params, creation, env_range):
        """Everything the constructor computes on the host before the simulator exists: the flat C config, the
        creation-time draws, the terrain.  No library call (tests drive the host build of the kernels with it)."""
        # ---- _parse_cfg (legged_robot.py:710-720)
        sim_dt = float(getattr(sim_params, "dt", cfg.sim.dt)) if sim_params is not None else float(cfg.sim.dt)
        self.dt = cfg.control.decimation * sim_dt
        self.obs_scales = cfg.normalization.obs_scales
        self.reward_scales = class_to_dict(cfg.rewards.scales)
        self.command_ranges = class_to_dict(cfg.commands.ranges)
        mesh_type = cfg.terrain.mesh_type
        if mesh_type not in ("plane", "heightfield", "trimesh"):
            raise ValueError("Terrain mesh type not recognised. Allowed types are [plane, heightfield, trimesh]")
        rough = mesh_type in ("heightfield", "trimesh")
        if getattr(cfg.env, "use_ref_actions", False):
            raise NotImplementedError("env.use_ref_actions=True (hector_env.py:159-160) is not built; no reference config sets it")
        if not rough:
            cfg.terrain.curriculum = False                   # legged_robot.py:714-716
        self.max_episode_length_s = cfg.env.episode_length_s
        self.max_episode_length = math.ceil(self.max_episode_length_s / self.dt)
        cfg.domain_rand.push_interval = math.ceil(cfg.domain_rand.push_interval_s / self.dt)
        self.total_envs = cfg.env.num_envs
        self.env_lo, self.env_hi = env_range if env_range is not None else (0, cfg.env.num_envs)
        self.num_envs = self.env_hi - self.env_lo
        self.num_obs = cfg.env.num_observations
        self.num_privileged_obs = cfg.env.num_privileged_obs
        self.num_actions = cfg.env.num_actions
        DOF_NAMES, BODY_NAMES, URDF_EFFORT = self.DOF_NAMES, self.BODY_NAMES, self.URDF_EFFORT
        nd = len(DOF_NAMES)
        self.num_dof = self.num_dofs = nd
        self.num_bodies = len(BODY_NAMES)
        self.dof_names, self.body_names = DOF_NAMES, BODY_NAMES
        self.obs_frame, self.priv_frame = 11 + 3 * nd, self.PRIV_BASE + 3 * nd         # 41 / 70, with arms 65 / 94, XBot-L 47 / 73
        self.obs_ld, self.priv_ld = -(-15 * self.obs_frame // 4) * 4, -(-self.PRIV_STACK * self.priv_frame // 4) * 4
        self.frame_dims = (self.obs_frame, self.priv_frame, 15, self.PRIV_STACK)      # what PPO.init_storage(frames=...) takes
        if (self.num_obs, self.num_privileged_obs, self.num_actions) != (15 * self.obs_frame, self.PRIV_STACK * self.priv_frame, nd):
            raise ValueError(f"{type(self).__name__} serves the {15 * self.obs_frame} / {self.PRIV_STACK * self.priv_frame} / {nd} layout only")
        self.feet_indices = [i for i, n in enumerate(BODY_NAMES) if cfg.asset.foot_name in n]
        self.knee_indices = [i for i, n in enumerate(BODY_NAMES) if cfg.asset.knee_name in n]
        self.termination_contact_indices = [i for k in cfg.asset.terminate_after_contacts_on
                                            for i, n in enumerate(BODY_NAMES) if k in n]
        self.penalised_contact_indices = [i for k in cfg.asset.penalize_contacts_on
                                          for i, n in enumerate(BODY_NAMES) if k in n]
        nl = nd // 2
        assert self.feet_indices == [self.FOOT_LOCAL, nl + self.FOOT_LOCAL] and self.knee_indices == [self.KNEE_LOCAL, nl + self.KNEE_LOCAL]
        assert sorted(self.penalised_contact_indices) == self._expected_penalised(nl)
        assert sorted(self.termination_contact_indices) == self._expected_termination(nd)

        # ---- create_sim (hector_env.py:114-133): terrain first, then the robots
        n = self.total_envs
        self.custom_origins = rough
        self.terrain = None
        terrain_grid = None
        if creation is not None:          # tests / shards: replay a recorded creation (friction, base mass, origins, start pose, terrain)
            friction, mass, start = (np.asarray(creation[k], np.float32) for k in ("friction", "mass", "start"))
            self.env_origins = np.asarray(creation["origins"], np.float32)
            if rough:
                terrain_grid = creation["terrain"]      # dict(heights int16 [R][C], horizontal_scale, vertical_scale, border_size)
                self._terrain_levels0, self.terrain_types = creation.get("terrain_levels"), creation.get("terrain_types")
                self.terrain_origins = creation.get("terrain_origins")
                self.max_terrain_level = cfg.terrain.num_rows
        else:
            if rough:
                from .terrain import HumanoidTerrain
                self.terrain = HumanoidTerrain(cfg.terrain, n)
                terrain_grid = dict(heights=self.terrain.heightsamples, horizontal_scale=cfg.terrain.horizontal_scale,
                                    vertical_scale=cfg.terrain.vertical_scale, border_size=cfg.terrain.border_size)
                self.env_origins = self._terrain_origins(cfg, n, self.terrain)
            else:
                self.env_origins = self._grid_origins(cfg, n)
            friction, mass, start = creation_randomisation(cfg, n, self.env_origins, self.BASE_MASS)
        self._terrain_grid = terrain_grid
        if env_range is not None:
            sl = slice(self.env_lo, self.env_hi)
            friction, mass, start, self.env_origins = friction[sl], mass[sl], start[sl], self.env_origins[sl]
        n = self.num_envs
        self.env_frictions, self.body_mass, self.start_pos = friction, mass, start

        # ---- flat C config
        c = capi.SimCfg()
        c.num_envs = n
        c.decimation = cfg.control.decimation
        c.sim_dt = sim_dt
        c.gravity_z = cfg.sim.gravity[2]
        c.action_scale = cfg.control.action_scale
        c.clip_actions = cfg.normalization.clip_actions
        c.clip_observations = cfg.normalization.clip_observations
        self.default_dof_pos = np.array([cfg.init_state.default_joint_angles[nm] for nm in DOF_NAMES], np.float32)
        self.p_gains, self.d_gains = np.zeros(nd, np.float32), np.zeros(nd, np.float32)
        for i, nm in enumerate(DOF_NAMES):           # substring match, legged_robot.py:486-500
            for key in cfg.control.stiffness:
                if key in nm:
                    self.p_gains[i] = cfg.control.stiffness[key]
                    self.d_gains[i] = cfg.control.damping[key]
        self.torque_limits = (np.array(URDF_EFFORT, np.float32) * np.float32(cfg.safety.torque_limit)).astype(np.float32)
        c.num_dof = nd
        for j in range(nd):
            c.default_dof_pos[j] = self.default_dof_pos[j]
            c.p_gains[j], c.d_gains[j] = self.p_gains[j], self.d_gains[j]
            c.torque_limits[j] = self.torque_limits[j]
        dr = cfg.domain_rand
        c.action_delay = getattr(dr, "action_delay", 0.0)
        c.action_noise = getattr(dr, "action_noise", 0.0)
        c.add_noise = int(cfg.noise.add_noise)
        c.noise_level = cfg.noise.noise_level
        ns, os_ = cfg.noise.noise_scales, self.obs_scales
        self.noise_scale_vec = self._noise_scale_vec(ns, os_)
        for k in range(self.obs_frame):
            c.noise_scale_vec[k] = self.noise_scale_vec[k]
        c.push_robots = int(dr.push_robots)
        c.push_interval = int(dr.push_interval)
        c.max_push_vel_xy = dr.max_push_vel_xy
        c.max_push_ang_vel = getattr(dr, "max_push_ang_vel", 0.0)
        c.resample_interval = int(cfg.commands.resampling_time / self.dt)
        c.heading_command = int(cfg.commands.heading_command)
        for i, key in enumerate(("lin_vel_x", "lin_vel_y", "ang_vel_yaw", "heading")):
            c.cmd_range[i][0], c.cmd_range[i][1] = self.command_ranges[key]
        c.obs_scale_lin_vel, c.obs_scale_ang_vel = os_.lin_vel, os_.ang_vel
        c.obs_scale_dof_pos, c.obs_scale_dof_vel, c.obs_scale_quat = os_.dof_pos, os_.dof_vel, os_.quat
        c.max_episode_length = float(self.max_episode_length)
        c.max_episode_length_s = float(self.max_episode_length_s)
        c.env_dt = self.dt
        base_init = cfg.init_state.pos + cfg.init_state.rot + cfg.init_state.lin_vel + cfg.init_state.ang_vel
        for k in range(13):
            c.base_init_state[k] = base_init[k]
        c.custom_origins = int(rough)
        # _prepare_reward_function (legged_robot.py:517-540): drop zero scales, multiply by dt
        unknown = [k for k, v in self.reward_scales.items() if v != 0 and k not in capi.REWARD_NAMES and k != "termination"]
        if unknown:
            raise ValueError(f"reward terms without a kernel implementation: {unknown}")
        self.reward_scales = {k: v * self.dt for k, v in self.reward_scales.items() if v != 0}
        self.reward_names = [k for k in self.reward_scales if k != "termination"]
        for i, nm in enumerate(capi.REWARD_NAMES):
            c.reward_scale[i] = self.reward_scales.get(nm, 0.0)
        rw = cfg.rewards
        c.only_positive_rewards = int(rw.only_positive_rewards)
        c.base_height_target, c.min_dist, c.max_dist = rw.base_height_target, rw.min_dist, rw.max_dist
        c.target_joint_pos_scale, c.target_feet_height = rw.target_joint_pos_scale, rw.target_feet_height
        c.cycle_time, c.tracking_sigma, c.max_contact_force = rw.cycle_time, rw.tracking_sigma, rw.max_contact_force
        # sim.physx of the config as far as the contact model has a place for it (include/hx_sim.h): hector_config.py:113-117
        px = cfg.sim.physx
        c.max_depenetration_velocity, c.contact_offset, c.rest_offset = float(px.max_depenetration_velocity), float(px.contact_offset), float(px.rest_offset)
        c.self_collisions = int(getattr(cfg.asset, "self_collisions", 1) == 0)      # the reference's bit filter: 0 = enabled
        for k, v in PHYS.items():      # model constants without a config counterpart; studies may also override the three above
            setattr(c, k, v)
        c.terrain_mu = cfg.terrain.static_friction
        c.env_id_offset = self.env_lo
        self._ccfg = c
        # slope_treshold of the trimesh conversion (utils/terrain.py:70-73): grid neighbours further apart in height than
        # slope_threshold * horizontal_scale are joined by a vertical wall; 'heightfield' has no walls
        self._wall = float(cfg.terrain.slope_treshold) * float(cfg.terrain.horizontal_scale) if (mesh_type == "trimesh" and getattr(cfg.terrain, "slope_treshold", None)) else 0.0
        return c, friction, mass, start, terrain_grid, rough

    def _create(self, L, c, friction, mass, start, terrain_grid, rough, stream, init_pack):
        cfg = self.cfg
        seed = int(getattr(cfg, "seed", 0)) & 0xFFFFFFFF
        h = capi.C.c_void_p()
        capi.check(L.hx_sim_create(capi.C.byref(c), capi.ptr(capi.farr(friction)), capi.ptr(capi.farr(mass)),
                                   capi.ptr(capi.farr(self.env_origins)), capi.ptr(capi.farr(start)),
                                   seed | (0x5EED << 32), stream, capi.C.byref(h)), "hx_sim_create")
        self._h = h
        self._L = L
        if rough:                         # _create_heightfield / _create_trimesh (legged_robot.py:553-585)
            hts = np.ascontiguousarray(terrain_grid["heights"], np.int16)
            self.height_samples = hts
            capi.check(L.hx_sim_set_terrain(h, hts.ctypes.data, hts.shape[0], hts.shape[1],
                                            float(terrain_grid["horizontal_scale"]), float(terrain_grid["vertical_scale"]),
                                            -float(terrain_grid["border_size"]), -float(terrain_grid["border_size"]), self._wall),
                       "hx_sim_set_terrain")
        self._curriculum = bool(rough and cfg.terrain.curriculum)
        if self._curriculum:              # _update_terrain_curriculum (legged_robot.py:399-419) runs inside the env-step kernel
            if getattr(self, "terrain_origins", None) is None or self._terrain_levels0 is None:
                raise ValueError("terrain curriculum needs terrain_origins / terrain_levels / terrain_types")
            og = np.ascontiguousarray(self.terrain_origins, np.float32)
            sl = slice(self.env_lo, self.env_hi)
            lv = np.ascontiguousarray(np.asarray(self._terrain_levels0)[sl], np.int32)
            ty = np.ascontiguousarray(np.asarray(self.terrain_types)[sl], np.int32)
            capi.check(L.hx_sim_set_terrain_curriculum(h, og.ctypes.data, og.shape[0], og.shape[1], lv.ctypes.data, ty.ctypes.data,
                                                       float(cfg.terrain.terrain_length), float(self.max_episode_length_s)),
                       "hx_sim_set_terrain_curriculum")
        self.stream = L.hx_sim_stream(h)
        self.common_step_counter = 0
        self.extras = {}
        self._keep = []
        # constructor tail: reset_idx(all) + compute_observations (hector_env.py:50-51)
        self._reset_all(init_pack)

    @staticmethod
    def _expected_penalised(nl):
        return [0, 3, nl + 3]                     # 'base', 'thigh' (hector_config.py:35)

    @staticmethod
    def _expected_termination(nd):
        return [0, 3, 8] if nd == 10 else [0, 3, 6, 7, 8, 12, 15, 16, 17]

    def _noise_scale_vec(self, ns, os_):
        """hector_env.py:135-155 (the last slice is [38:42] on a 41-vector there)"""
        v = np.zeros(self.obs_frame, np.float32)
        v[5:15] = ns.dof_pos * os_.dof_pos
        v[15:25] = ns.dof_vel * os_.dof_vel
        v[35:38] = ns.ang_vel * os_.ang_vel
        v[38:41] = ns.quat * os_.quat
        return v

    def _terrain_origins(self, cfg, n, terrain):
        """legged_robot.py:687-697: a random level per robot, tile type by robot index."""
        max_init_level = cfg.terrain.max_init_terrain_level
        if not cfg.terrain.curriculum:
            max_init_level = cfg.terrain.num_rows - 1
        try:
            import torch
            levels = torch.randint(0, max_init_level + 1, (n,)).numpy()
        except ImportError:                                   # pragma: no cover
            levels = np.random.randint(0, max_init_level + 1, n)
        self._terrain_levels0 = levels.astype(np.int64)
        self.terrain_types = np.floor(np.arange(n) / (n / cfg.terrain.num_cols)).astype(np.int64)
        self.max_terrain_level = cfg.terrain.num_rows
        self.terrain_origins = terrain.env_origins.astype(np.float32)
        return self.terrain_origins[self._terrain_levels0, self.terrain_types].copy()

    @staticmethod
    def _grid_origins(cfg, n):
        num_cols = int(np.floor(np.sqrt(n)))
        num_rows = int(np.ceil(n / num_cols))
        xx, yy = np.meshgrid(np.arange(num_rows), np.arange(num_cols), indexing="ij")
        origins = np.zeros((n, 3), np.float32)
        origins[:, 0] = cfg.env.env_spacing * xx.flatten()[:n]
        origins[:, 1] = cfg.env.env_spacing * yy.flatten()[:n]
        return origins

    # ------------------------------------------------------------------ buffers
    def _buf(self, which, shape, dtype=np.float32, strides=None):
        p = capi.C.c_void_p()
        capi.check(self._L.hx_sim_buffer(self._h, which, capi.C.byref(p)), "hx_sim_buffer")
        return DeviceArray(p.value, shape, dtype, strides, self.stream, owner=self)

    def _refresh_views(self):
        n = self.num_envs
        self.obs_buf = self._buf(capi.BUF_OBS, (n, self.num_obs), strides=(self.obs_ld * 4, 4))
        self.privileged_obs_buf = self._buf(capi.BUF_PRIV, (n, self.num_privileged_obs), strides=(self.priv_ld * 4, 4))
        self.rew_buf = self._buf(capi.BUF_REW, (n,))
        self.reset_buf = self._buf(capi.BUF_RESET, (n,), np.uint8)
        self.time_out_buf = self._buf(capi.BUF_TIMEOUT, (n,), np.uint8)
        self.extras["time_outs"] = self._buf(capi.BUF_TIMEOUT_VISIBLE, (n,), np.uint8)

    @property
    def episode_length_buf(self):
        return self._buf(capi.BUF_EP_LEN, (self.num_envs,), np.int32)

    @episode_length_buf.setter
    def episode_length_buf(self, value):
        """`env.episode_length_buf = randint_like(...)` of on_policy_runner.py:103-106."""
        arr = value.numpy() if isinstance(value, DeviceArray) else (value.cpu().numpy() if hasattr(value, "cpu") else np.asarray(value))
        arr = np.ascontiguousarray(arr, np.int32)
        capi.check(self._L.hx_sim_set_episode_length(self._h, capi.ptr(arr)), "set_episode_length")

    def _reset_all(self, pack):
        p = None
        if pack is not None:
            p, keep = device_pointer(np.ascontiguousarray(pack, np.float32))
            self._keep = [keep]
        capi.check(self._L.hx_sim_reset_all(self._h, p), "hx_sim_reset_all")
        self._refresh_views()

    # ------------------------------------------------------------------ VecEnv
    def step(self, actions, pack=None):
        a_ptr, keep_a = device_pointer(actions)
        p_ptr, keep_p = (None, None) if pack is None else device_pointer(np.ascontiguousarray(pack, np.float32))
        self._keep = [keep_a, keep_p]
        capi.check(self._L.hx_sim_step(self._h, a_ptr, p_ptr), "hx_sim_step")
        self.common_step_counter += 1
        self._refresh_views()
        return self.obs_buf, self.privileged_obs_buf, self.rew_buf, self.reset_buf, self.extras

    def reset(self):
        """reset all robots, then one zero-action step (legged_robot.py:111-116)."""
        # the library's constructor-style reset does not re-randomise the creation pose; mirror reset_idx(all)
        self._reset_all(None)
        zeros = capi.DeviceBuffer(self.num_envs * self.num_actions * 4)
        obs, priv, _, _, _ = self.step(DeviceArray(zeros.ptr, (self.num_envs, self.num_actions), owner=zeros))
        capi.check(self._L.hx_sync(self.stream), "sync")
        return obs, priv

    def get_observations(self):
        return self.obs_buf

    def get_privileged_observations(self):
        return self.privileged_obs_buf

    # ------------------------------------------------------------------ state access (tests, play-style scripts)
    def get_state(self):
        n = self.num_envs
        root, q, qd = np.empty((n, 13), np.float32), np.empty((n, self.num_dof), np.float32), np.empty((n, self.num_dof), np.float32)
        capi.check(self._L.hx_sim_get_state(self._h, capi.ptr(root), capi.ptr(q), capi.ptr(qd)), "get_state")
        return root, q, qd

    def set_state(self, root, q, qd):
        capi.check(self._L.hx_sim_set_state(self._h, capi.ptr(capi.farr(root)), capi.ptr(capi.farr(q)), capi.ptr(capi.farr(qd))), "set_state")

    def set_step_counter(self, c):
        self.common_step_counter = int(c)
        capi.check(self._L.hx_sim_set_step_counter(self._h, int(c)), "set_step_counter")

    @property
    def root_states(self):
        return self.get_state()[0]

    @property
    def dof_pos(self):
        return self.get_state()[1]

    @property
    def dof_vel(self):
        return self.get_state()[2]

    @property
    def commands(self):
        return self._buf(capi.BUF_COMMANDS, (4, self.num_envs)).numpy().T

    @commands.setter
    def commands(self, value):
        """`env.commands[:, 0] = 0.5` of the reference (play.py:136-140) becomes read-modify-assign:
        c = env.commands; c[:, 0] = 0.5; env.commands = c."""
        capi.check(self._L.hx_sim_set_commands(self._h, capi.ptr(capi.farr(np.asarray(value, np.float32).reshape(self.num_envs, 4)))),
                   "hx_sim_set_commands")

    def _base_velocities(self):
        lin, ang = np.empty((self.num_envs, 3), np.float32), np.empty((self.num_envs, 3), np.float32)
        capi.check(self._L.hx_sim_get_base_velocities(self._h, capi.ptr(lin), capi.ptr(ang)), "hx_sim_get_base_velocities")
        return lin, ang

    @property
    def base_lin_vel(self):
        return self._base_velocities()[0]

    @property
    def base_ang_vel(self):
        return self._base_velocities()[1]

    @property
    def torques(self):
        return self._buf(capi.BUF_TORQUES, (self.num_dof, self.num_envs)).numpy().T

    @property
    def contact_forces(self):
        return self._buf(capi.BUF_CONTACT, (self.num_bodies, 3, self.num_envs)).numpy().transpose(2, 0, 1)

    def episode_stats(self):
        """extras['episode'] of legged_robot.py:198-201 averaged over the envs that reset since the last call,
        plus the runner's Train/mean_reward and Train/mean_episode_length over the same episodes."""
        mean = np.zeros(capi.NUM_REWARDS + 2, np.float32)
        cnt = capi.C.c_int32(0)
        capi.check(self._L.hx_sim_episode_stats(self._h, capi.ptr(mean), capi.C.byref(cnt)), "episode_stats")
        info = {"rew_" + k: float(mean[capi.REWARD_NAMES.index(k)]) for k in self.reward_names}
        if self.cfg.terrain.mesh_type == "trimesh" and getattr(self, "_terrain_levels0", None) is not None:
            info["terrain_level"] = float(np.mean(self.terrain_levels))           # legged_robot.py:203-204
        self.last_episode_return, self.last_episode_length = float(mean[capi.NUM_REWARDS]), float(mean[capi.NUM_REWARDS + 1])
        return info, cnt.value

    @property
    def terrain_levels(self):
        """legged_robot.py:693 / :414-418: the row of the tile map each robot is on (moves with the curriculum)."""
        if getattr(self, "_terrain_levels0", None) is None:
            return None
        if not getattr(self, "_curriculum", False):
            return np.asarray(self._terrain_levels0)[getattr(self, "env_lo", 0):getattr(self, "env_hi", None)]
        lv = np.zeros(self.num_envs, np.int32)
        capi.check(self._L.hx_sim_get_terrain_levels(self._h, lv.ctypes.data), "hx_sim_get_terrain_levels")
        return lv.astype(np.int64)

    def sync(self):
        capi.check(self._L.hx_sync(self.stream), "sync")

    def close(self):
        if getattr(self, "_h", None):
            self._L.hx_sim_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PipelinedHectorEnv(VecEnv):
    """The same batch of robots as HectorFreeEnv(cfg), run as `num_shards` independent simulators on their own HIP
    streams.  The env-step kernel is latency-bound (eight lanes per robot: 512 waves on 1024 SIMDs) while the
    policy GEMMs are throughput-bound, so when the runner drives the shards round-robin
    (OnPolicyRunner.learn) one shard's physics overlaps the other shards' GEMMs on the same GPU.
    Robots, random streams (Philox keyed by global env id) and results are those of the unsharded env.
    One documented difference: the reference's stale-`extras["time_outs"]` quirk is
    evaluated per shard ("some env of this shard reset this step") instead of over the whole batch."""

    def __init__(self, cfg, sim_params=None, physics_engine=None, sim_device="cuda:0", headless=True, num_shards=2):
        n = cfg.env.num_envs
        assert n % num_shards == 0
        per = n // num_shards
        # draw the creation-time randomisation ONCE for the whole batch, in the reference's order
        creation = {}
        if cfg.terrain.mesh_type in ("heightfield", "trimesh"):
            from .terrain import HumanoidTerrain
            self.terrain = HumanoidTerrain(cfg.terrain, n)
            creation["terrain"] = dict(heights=self.terrain.heightsamples, horizontal_scale=cfg.terrain.horizontal_scale,
                                       vertical_scale=cfg.terrain.vertical_scale, border_size=cfg.terrain.border_size)
            probe = HectorFreeEnv.__new__(HectorFreeEnv)
            probe_origins = probe._terrain_origins(cfg, n, self.terrain)
            creation["terrain_levels"], creation["terrain_types"] = probe._terrain_levels0, probe.terrain_types
            creation["terrain_origins"] = probe.terrain_origins
        else:
            probe_origins = HectorFreeEnv._grid_origins(cfg, n)
        friction, mass, start = creation_randomisation(cfg, n, probe_origins)
        creation.update(friction=friction, mass=mass, start=start, origins=probe_origins)
        self.shards = [HectorFreeEnv(cfg, sim_params, physics_engine, sim_device, headless, creation=creation,
                                     env_range=(i * per, (i + 1) * per)) for i in range(num_shards)]
        s0 = self.shards[0]
        self.cfg, self.num_envs, self.num_obs, self.num_privileged_obs, self.num_actions = cfg, n, s0.num_obs, s0.num_privileged_obs, s0.num_actions
        self.obs_ld, self.priv_ld = s0.obs_ld, s0.priv_ld
        self.max_episode_length, self.dt, self.device = s0.max_episode_length, s0.dt, s0.device
        self.reward_names = s0.reward_names
        self.stream = s0.stream
        self.extras = {}

    @property
    def episode_length_buf(self):
        return np.concatenate([s.episode_length_buf.numpy() for s in self.shards])

    @episode_length_buf.setter
    def episode_length_buf(self, value):
        arr = value.cpu().numpy() if hasattr(value, "cpu") else np.asarray(value)
        for s in self.shards:
            s.episode_length_buf = arr[s.env_lo:s.env_hi]

    def step(self, actions):
        """Whole-batch step (no overlap); the runner uses the shards directly instead."""
        from ..devarray import DeviceArray
        p, keep = device_pointer(actions)
        outs = []
        for s in self.shards:
            a = DeviceArray(p + s.env_lo * self.num_actions * 4, (s.num_envs, self.num_actions), owner=keep)
            outs.append(s.step(a))
        self.sync()
        return outs

    def reset(self):
        return [s.reset() for s in self.shards]

    def get_observations(self):
        return [s.get_observations() for s in self.shards]

    def get_privileged_observations(self):
        return [s.get_privileged_observations() for s in self.shards]

    def episode_stats(self):
        infos, cnts = zip(*[s.episode_stats() for s in self.shards])
        tot = sum(cnts)
        w = [c / tot if tot else 0.0 for c in cnts]
        info = {k: sum(wi * i[k] for wi, i in zip(w, infos)) for k in infos[0]}
        if "terrain_level" in info:                    # a mean over ALL robots (legged_robot.py:203-204), not over the episodes
            info["terrain_level"] = sum(s.num_envs * i["terrain_level"] for s, i in zip(self.shards, infos)) / self.num_envs
        self.last_episode_return = sum(wi * s.last_episode_return for wi, s in zip(w, self.shards))
        self.last_episode_length = sum(wi * s.last_episode_length for wi, s in zip(w, self.shards))
        return info, tot

    def sync(self):
        for s in self.shards:
            s.sync()

    def close(self):
        for s in self.shards:
            s.close()


class HectorFullFreeEnv(HectorFreeEnv):
    """Task `hector_full` (reference humanoid/envs/custom/hector_w_arm_env.py HectorFullFreeEnv): the same biped with its
    two 4-joint arms actuated, 18 DoF in Isaac Gym's order (L leg, L arm, R leg, R arm), observation frames 65 / 94 wide.
    Same kernel source as hector, instantiated with the arm chains (hx_sim_cfg.num_dof = 18)."""
    DOF_NAMES = ["L_hip_joint", "L_hip_roll_joint", "L_thigh_joint", "L_calf_joint", "L_toe_joint",
                 "L_shoulder_yaw_joint", "L_shoulder_pitch_joint", "L_shoulder_roll_joint", "L_elbow_joint",
                 "R_hip_joint", "R_hip_roll_joint", "R_thigh_joint", "R_calf_joint", "R_toe_joint",
                 "R_shoulder_yaw_joint", "R_shoulder_pitch_joint", "R_shoulder_roll_joint", "R_elbow_joint"]
    BODY_NAMES = ["base", "L_hip", "L_hip2", "L_thigh", "L_calf", "L_toe", "L_twist", "L_shoulder", "L_roll", "L_elbow",
                  "R_hip", "R_hip2", "R_thigh", "R_calf", "R_toe", "R_twist", "R_shoulder", "R_roll", "R_elbow"]
    # robot_w_arm.urdf <limit effort=...>: the right elbow carries 24 where the other arm joints carry 17
    URDF_EFFORT = [33.5, 33.5, 33.5, 67.0, 33.5, 17.0, 17.0, 17.0, 17.0, 33.5, 33.5, 33.5, 67.0, 33.5, 17.0, 17.0, 17.0, 24.0]
    BASE_MASS = 4.982                                   # collapsed base link of robot_w_arm.urdf (tools/compile_urdf.py --full)

    def _noise_scale_vec(self, ns, os_):
        """hector_w_arm_env.py:157-161, overlapping slices included (index 58 ends up with the angular-velocity scale)"""
        v = np.zeros(self.obs_frame, np.float32)
        v[5:23] = ns.dof_pos * os_.dof_pos
        v[23:41] = ns.dof_vel * os_.dof_vel
        v[41:59] = 0.0
        v[58:61] = ns.ang_vel * os_.ang_vel
        v[61:65] = ns.quat * os_.quat
        return v


class XBotLFreeEnv(HectorFreeEnv):
    """Task `humanoid_ppo` (reference humanoid/envs/custom/humanoid_env.py XBotLFreeEnv + humanoid_config.py XBotLCfg): the
    12-DoF XBot-L humanoid, whose joints turn about the z axes of rotated joint frames.  Same kernel source, instantiated with
    the XBot model descriptor (hx_sim_cfg.num_dof = 12): observation frame 47 wide x 15, its own privileged frame of 73 values
    x c_frame_stack = 3, gait reference on joints 2-4 / 8-10, only base_link terminates an episode."""
    DOF_NAMES = [f"{s}_{j}_joint" for s in ("left", "right") for j in ("leg_roll", "leg_yaw", "leg_pitch", "knee", "ankle_pitch", "ankle_roll")]
    BODY_NAMES = ["base_link"] + [f"{s}_{j}_link" for s in ("left", "right") for j in ("leg_roll", "leg_yaw", "leg_pitch", "knee", "ankle_pitch", "ankle_roll")]
    URDF_EFFORT = [100.0, 100.0, 250.0, 250.0, 100.0, 100.0] * 2          # XBot-L.urdf <limit effort=...>
    BASE_MASS = 29.900618661923257                                       # collapsed base link (tools/compile_urdf.py --xbot)
    PRIV_BASE, PRIV_STACK = 37, 3
    KNEE_LOCAL, FOOT_LOCAL = 4, 6

    @staticmethod
    def _expected_penalised(nl):
        return [0]                                # 'base_link' (humanoid_config.py:67-68)

    @staticmethod
    def _expected_termination(nd):
        return [0]

    def _noise_scale_vec(self, ns, os_):
        """humanoid_env.py:179-186"""
        v = np.zeros(self.obs_frame, np.float32)
        v[5:17] = ns.dof_pos * os_.dof_pos
        v[17:29] = ns.dof_vel * os_.dof_vel
        v[41:44] = ns.ang_vel * os_.ang_vel
        v[44:47] = ns.quat * os_.quat
        return v
