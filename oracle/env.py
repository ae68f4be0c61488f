"""ORACLE (test infrastructure, never shipped) -- CPU restatement of the hector env step.

Follows, function by function, the reference's
  humanoid/envs/custom/hector_env.py   (step :158-169, compute_observations :172-254, _push_robots :53-68,
                                        _get_gait_phase :75-88, reset_idx :256-261, rewards :277-539)
  humanoid/envs/base/legged_robot.py   (step :84-108, post_physics_step :118-153, check_termination :155-160,
                                        reset_idx :162-214, compute_reward :216-234, _resample_commands :321-335,
                                        _compute_torques :339-355, _reset_dofs :358-372, _reset_root_states :373-396,
                                        _post_physics_step_callback :303-319)
in float32 numpy.  The rigid-body step underneath is oracle/physics.py (float64) -- see that file's
header for why the dynamics are "parity unpinned".

Pinning: tests/test_oracle_env.py replays tests/golden/env_rollout_*.npz, which were produced by the
reference's own code (tests/golden/make_env_fixtures.py), and requires this restatement to reproduce
the reference's observations / rewards / resets / time-outs step for step.

Random numbers are injected ("random pack", include/hx_sim.h HX_RP_*): pack[field, env].
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np

from . import physics as P

F = np.float32
RP = dict(delay=0, act_noise=1, cmd_a=11, push=14, reset_q=19, reset_xy=29, cmd_b=31, obs_noise=34, level=75)
RP_SIZE = 75

# constants of HectorCfg (reference hector_config.py); the product reads them from its own config classes
DEFAULT_Q = np.array([0, 0, .785, -1.578, .785] * 2, F)
KP = np.array([40, 40, 60, 120, 20] * 2, F)
KD = np.array([3, 3, 5, 4, 1] * 2, F)
EFFORT = np.array([33.5, 33.5, 33.5, 67, 33.5] * 2, F)
TORQUE_LIMIT = (EFFORT * F(0.85)).astype(F)          # legged_robot.py:292, hector_config.py:26
FEET = [5, 10]
KNEES = [4, 9]
TERM = [0, 3, 8]
REWARD_ORDER = ["action_smoothness", "base_acc", "base_height", "collision", "default_joint_pos", "dof_acc",
                "dof_vel", "feet_air_time", "feet_clearance", "feet_contact_forces", "feet_contact_number",
                "feet_distance", "foot_slip", "knee_distance", "orientation", "torques", "tracking_ang_vel",
                "tracking_lin_vel"]
REWARD_SCALE = dict(action_smoothness=-0.008, base_acc=0.3, base_height=1.0, collision=-0.5,
                    default_joint_pos=1.7, dof_acc=-1e-6, dof_vel=-1e-4, feet_air_time=2.0, feet_clearance=1.5,
                    feet_contact_forces=-0.05, feet_contact_number=2.5, feet_distance=0.2, foot_slip=-0.05,
                    knee_distance=0.2, orientation=2.0, torques=-1e-5, tracking_ang_vel=1.5, tracking_lin_vel=2.5)
NOISE_VEC = np.zeros(41, F)
NOISE_VEC[5:15] = 0.05 * 1.0
NOISE_VEC[15:25] = 0.5 * 0.05
NOISE_VEC[35:38] = 0.1 * 1.0
NOISE_VEC[38:41] = 0.03 * 1.0      # reference writes [38:42] on a 41-vector (hector_env.py:154)


def rp_layout(ndof, nobs):
    """Random-pack field offsets (include/hx_sim.h HX_RP_* for the hector task): delay 1, action noise ndof, command
    resample 3, push 5, reset dofs ndof, reset xy 2, command resample at reset 3, observation noise nobs, terrain level 1."""
    o, out = 0, {}
    for name, w in (("delay", 1), ("act_noise", ndof), ("cmd_a", 3), ("push", 5), ("reset_q", ndof), ("reset_xy", 2),
                    ("cmd_b", 3), ("obs_noise", nobs), ("level", 1)):
        out[name] = o
        o += w
    return out


class Task:
    """What distinguishes one task of the hector family from another in the env glue (everything else is shared code)."""

    def __init__(self, name, model_json, default_q, kp, kd, torque_limit, feet, knees, term, penal, noise_vec, reward_scale,
                 opts, max_contact_force, min_dist, djp_pairs, djp_arm_pairs=(), priv_base=40, priv_stack=15, ref_right=7,
                 base_init_z=0.55, base_height_target=0.55, clip=100.0, xbot_priv=False):
        self.name, self.model_json = name, model_json
        self.default_q, self.kp, self.kd, self.torque_limit = (np.asarray(x, F) for x in (default_q, kp, kd, torque_limit))
        self.ndof = len(self.default_q)
        self.nobs, self.npriv = 11 + 3 * self.ndof, priv_base + 3 * self.ndof   # 41 / 70 for 10 DoF, 65 / 94 for 18, 47 / 73 XBot-L
        # frames per privileged row (c_frame_stack); first right-leg joint of the gait reference (7: hector_env.py:105-107 and
        # hector_w_arm_env.py:107-114; 8: humanoid_env.py:136-138); init_state.pos z; rewards.base_height_target; normalization clips;
        # whether the privileged frame is humanoid_env.py:218-236's (no foot / root positions, with q - ref_dof_pos)
        self.priv_stack, self.ref_right, self.base_init_z, self.base_height_target = priv_stack, ref_right, base_init_z, base_height_target
        self.clip, self.xbot_priv = clip, xbot_priv
        self.feet, self.knees, self.term, self.penal = feet, knees, term, penal
        self.noise_vec = np.asarray(noise_vec, F)
        self.reward_scale, self.opts = dict(reward_scale), dict(opts)
        self.max_contact_force, self.min_dist = max_contact_force, min_dist
        self.djp_pairs, self.djp_arm_pairs = djp_pairs, djp_arm_pairs       # joint slices of _reward_default_joint_pos
        self.rp = rp_layout(self.ndof, self.nobs)
        self.rp_size = self.rp["level"]


_OPTS = dict(heading_command=True, only_positive_rewards=True, push_robots=True,
             cmd_ranges=dict(lin_vel_x=(-0.6, 0.6), lin_vel_y=(-0.3, 0.3), ang_vel_yaw=(-0.3, 0.3), heading=(-3.14, 3.14)),
             max_push_vel_xy=0.3, max_push_ang_vel=0.4, action_delay=0.0, action_noise=0.02)
HECTOR = Task("hector", P.MODEL_JSON, DEFAULT_Q, KP, KD, TORQUE_LIMIT, FEET, KNEES, TERM, TERM, NOISE_VEC, REWARD_SCALE, _OPTS,
              max_contact_force=180.0, min_dist=0.1, djp_pairs=((0, 2), (5, 7)))


def _hector_full():
    """hector with arms (reference hector_w_arm_config.py / hector_w_arm_env.py; DoF order L leg, L arm, R leg, R arm)."""
    leg_q, arm_q = [0, 0, .785, -1.578, .785], [0, 0, 0, -.785]
    leg_kp, arm_kp, leg_kd, arm_kd = [80, 80, 80, 80, 60], [30] * 4, [5, 5, 5, 5, 3], [3] * 4       # :97-100
    # URDF efforts as compiled (the right elbow carries 24 Nm where the other arm joints carry 17: robot_w_arm.urdf)
    eff = np.array([b["effort"] for b in P.load_model(P.MODEL_FULL_JSON)["bodies"][1:]], F)
    side = lambda leg, arm: leg + arm
    nv = np.zeros(65, F)                                   # hector_w_arm_env.py:157-161, overlapping slices included
    nv[5:23] = 0.05 * 1.0
    nv[23:41] = 0.5 * 0.05
    nv[41:59] = 0.0
    nv[58:61] = 0.1 * 1.0
    nv[61:65] = 0.03 * 1.0
    scale = dict(action_smoothness=-0.002, base_acc=0.22, base_height=0.8, collision=-1.0, default_joint_pos=1.2, dof_acc=-1e-6,
                 dof_vel=-1e-3, feet_air_time=1.5, feet_clearance=1.2, feet_contact_forces=-0.02, feet_contact_number=1.5,
                 feet_distance=0.2, foot_slip=-0.05, knee_distance=0.2, low_speed=0.2, orientation=1.0, torques=-1e-5,
                 track_vel_hard=0.5, tracking_ang_vel=1.1, tracking_lin_vel=1.2, vel_mismatch_exp=0.5)    # :165-193
    opts = dict(_OPTS)
    opts["cmd_ranges"] = dict(_OPTS["cmd_ranges"], lin_vel_x=(-0.6, 0.8))                               # :146
    opts["max_push_vel_xy"] = 0.5                                                                       # :133
    return Task("hector_full", P.MODEL_FULL_JSON, side(leg_q, arm_q) * 2, side(leg_kp, arm_kp) * 2, side(leg_kd, arm_kd) * 2,
                eff * F(0.85), feet=[5, 14], knees=[4, 13],
                term=[0, 3, 12, 7, 16, 6, 15, 8, 17],      # 'base','thigh','shoulder','twist','roll' in that order (:35)
                penal=[0, 3, 12], noise_vec=nv, reward_scale=scale, opts=opts, max_contact_force=200.0, min_dist=0.2,
                djp_pairs=((0, 2), (9, 11)), djp_arm_pairs=((5, 7), (14, 16)))


HECTOR_FULL = _hector_full()


def _humanoid():
    """XBot-L (reference humanoid_config.py XBotLCfg / humanoid_env.py XBotLFreeEnv; task humanoid_ppo)."""
    names = ["leg_roll", "leg_yaw", "leg_pitch", "knee", "ankle_pitch", "ankle_roll"]
    kp = [{"leg_roll": 200.0, "leg_yaw": 200.0, "leg_pitch": 350.0, "knee": 350.0, "ankle": 15.0}[k] for k in
          ("leg_roll", "leg_yaw", "leg_pitch", "knee", "ankle", "ankle")]                                   # :99-102, substring match
    eff = np.array([b["effort"] for b in P.load_model(P.MODEL_XBOT_JSON)["bodies"][1:]], F)
    nv = np.zeros(47, F)                                     # humanoid_env.py:179-186
    nv[5:17] = 0.05 * 1.0
    nv[17:29] = 0.5 * 0.05
    nv[41:44] = 0.1 * 1.0
    nv[44:47] = 0.03 * 1.0
    scale = dict(joint_pos=1.6, feet_clearance=1.0, feet_contact_number=1.2, feet_air_time=1.0, foot_slip=-0.05, feet_distance=0.2,
                 knee_distance=0.2, feet_contact_forces=-0.01, tracking_lin_vel=1.2, tracking_ang_vel=1.1, vel_mismatch_exp=0.5,
                 low_speed=0.2, track_vel_hard=0.5, default_joint_pos=0.5, orientation=1.0, base_height=0.2, base_acc=0.2,
                 action_smoothness=-0.002, torques=-1e-5, dof_vel=-5e-4, dof_acc=-1e-7, collision=-1.0)           # :188-218
    opts = dict(_OPTS)
    opts["cmd_ranges"] = dict(_OPTS["cmd_ranges"], lin_vel_x=(-0.3, 0.6))                                      # :173
    opts.update(max_push_vel_xy=0.2, action_delay=0.5)                                                         # :155-159
    assert len(names) == 6
    return Task("humanoid_ppo", P.MODEL_XBOT_JSON, [0.0] * 12, kp * 2, [10.0] * 12, eff * F(0.85), feet=[6, 12], knees=[4, 10],
                term=[0], penal=[0], noise_vec=nv, reward_scale=scale, opts=opts, max_contact_force=700.0, min_dist=0.2,
                djp_pairs=((0, 2), (6, 8)), priv_base=37, priv_stack=3, ref_right=8, base_init_z=0.95, base_height_target=0.89,
                clip=18.0, xbot_priv=True)


HUMANOID = _humanoid()


def quat_rotate_inverse(q, v):
    qw = q[:, 3:4]
    qv = q[:, :3]
    a = v * (F(2.0) * qw * qw - F(1.0))
    b = np.cross(qv, v) * qw * F(2.0)
    c = qv * np.sum(qv * v, 1, keepdims=True) * F(2.0)
    return (a - b + c).astype(F)


def quat_apply(q, v):
    xyz = q[:, :3]
    t = np.cross(xyz, v) * F(2)
    return (v + q[:, 3:4] * t + np.cross(xyz, t)).astype(F)


def euler_xyz_wrapped(q):
    """get_euler_xyz (isaacgym.torch_utils) then the (-pi,pi] wrap of legged_robot.py:50-55."""
    qx, qy, qz, qw = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    roll = np.arctan2(F(2) * (qw * qx + qy * qz), qw * qw - qx * qx - qy * qy + qz * qz)
    sinp = F(2) * (qw * qy - qz * qx)
    pitch = np.where(np.abs(sinp) >= 1, np.sign(sinp) * F(np.pi / 2), np.arcsin(np.clip(sinp, -1, 1)))
    yaw = np.arctan2(F(2) * (qw * qz + qx * qy), qw * qw + qx * qx - qy * qy - qz * qz)
    e = np.stack([roll, pitch, yaw], 1).astype(F) % F(2 * np.pi)
    e = np.where(e > F(np.pi), e - F(2 * np.pi), e)
    return e.astype(F)


def wrap_to_pi(a):
    a = a % F(2 * np.pi)
    return (a - F(2 * np.pi) * (a > F(np.pi))).astype(F)


def unif(lo, hi, u):
    return (F(hi - lo) * u + F(lo)).astype(F)


class HectorEnvOracle:
    def __init__(self, n, shape_friction, base_mass, env_origins, init_pack, add_noise=True,
                 start_xy=None, phys_dtype=np.float64, terrain=None, custom_origins=False, curriculum=None,
                 reward_scales=None, opts=None, task=HECTOR):
        """terrain: oracle.terrain.HeightField or None (plane).  custom_origins: True for heightfield/trimesh
        (legged_robot.py:688), which adds U[-1,1] to the reset xy (:381-384).
        curriculum: None or dict(origins [rows][cols][3], levels [n], types [n], env_length) -- the terrain curriculum of
        legged_robot.py:399-419; packs then carry one more row, RP["level"]."""
        self.n = n
        self.task = T = task
        # reward_scales: overrides / additions to HectorCfg's scales (hector_config.py:161-189), e.g. the four terms HectorCfg
        # zero-scales (joint_pos, low_speed, track_vel_hard, vel_mismatch_exp).  Active terms are evaluated in alphabetical
        # order like the reference (dir(), helpers.py:47; zero scales are dropped, legged_robot.py:521-527).
        sc = dict(T.reward_scale)
        sc.update(reward_scales or {})
        self.reward_scale = {k: v for k, v in sc.items() if v != 0}
        self.reward_order = sorted(self.reward_scale)
        self.ref_dof_pos = np.zeros((n, T.ndof), F)
        # opts: config switches / ranges that differ from HectorCfg (all keys optional):
        #   heading_command (commands.heading_command), only_positive_rewards, push_robots, cmd_ranges = dict(lin_vel_x=,
        #   lin_vel_y=, ang_vel_yaw=, heading=), max_push_vel_xy, max_push_ang_vel, action_delay, action_noise
        o = dict(T.opts)
        o["cmd_ranges"] = dict(T.opts["cmd_ranges"])
        for k, v in (opts or {}).items():
            if k == "cmd_ranges":
                o["cmd_ranges"].update(v)
            else:
                o[k] = v
        self.opts = o
        self.custom_origins = custom_origins
        self.curriculum = curriculum
        self.init_done = False
        if curriculum is not None:
            self.terrain_origins = np.asarray(curriculum["origins"], F)
            self.terrain_levels = np.asarray(curriculum["levels"], np.int64).copy()
            self.terrain_types = np.asarray(curriculum["types"], np.int64)
            self.max_terrain_level = self.terrain_origins.shape[0]
            self.env_length = float(curriculum["env_length"])
        model = P.load_model(T.model_json)
        m0 = model["bodies"][0]["mass"]
        self.phys = P.HectorPhysics(n, base_mass_added=np.asarray(base_mass, np.float64) - m0,
                                    shape_friction=shape_friction, dtype=phys_dtype, terrain=terrain, model=model)
        if T.name == "humanoid_ppo":                        # asset.self_collisions = 0 (humanoid_config.py:66)
            self.phys.enable_self_collision()
        self.state = P.State(n, phys_dtype, T.ndof)
        if start_xy is not None:
            # actor creation pose (legged_robot.py:653-655): origin + U[-1,1]^2, z of the origin; the first
            # privileged frames read body poses from this pose because reset does not refresh them
            self.state.root_pos[:] = np.asarray(start_xy, phys_dtype)
        self.env_origins = np.array(env_origins, F)          # own copy: the terrain curriculum re-bases rows
        self.env_frictions = np.asarray(shape_friction, F).reshape(n, 1)
        self.body_mass = np.asarray(base_mass, F).reshape(n, 1)
        self.add_noise = add_noise
        self.dt = 0.01
        self.max_episode_length = 2400.0
        z = lambda *s: np.zeros(s, F)
        self.actions, self.last_actions, self.last_last_actions = z(n, T.ndof), z(n, T.ndof), z(n, T.ndof)
        self.last_dof_vel, self.last_root_vel = z(n, T.ndof), z(n, 6)
        self.commands = z(n, 4)
        self.feet_air_time = z(n, 2)
        self.last_contacts = np.zeros((n, 2), bool)
        self.feet_height = z(n, 2)
        self.last_feet_z = F(0.05) * np.ones((n, 2), F)      # scalar 0.05 in the reference (hector_env.py:48)
        self.rand_push_force, self.rand_push_torque = z(n, 3), z(n, 3)
        self.episode_length_buf = np.zeros(n, np.int64)
        self.common_step_counter = 0
        self.reset_buf = np.ones(n, bool)
        self.time_out_buf = np.zeros(n, bool)
        self.time_outs_visible = np.zeros(n, bool)      # extras["time_outs"], stale unless some env reset
        self.episode_sums = {k: z(n) for k in self.reward_order}
        self.rew_buf = z(n)
        self.torques = z(n, T.ndof)
        self.obs_hist = z(15, n, T.nobs)       # oldest .. newest
        self.priv_hist = z(T.priv_stack, n, T.npriv)
        self.extras_episode = {}
        # tensors the glue reads from the simulator
        self._refresh(full=True)
        self.base_lin_vel = quat_rotate_inverse(self.root[:, 3:7], self.root[:, 7:10])
        self.base_ang_vel = quat_rotate_inverse(self.root[:, 3:7], self.root[:, 10:13])
        self.projected_gravity = quat_rotate_inverse(self.root[:, 3:7], np.tile(np.array([[0, 0, -1]], F), (n, 1)))
        self.base_euler = euler_xyz_wrapped(self.root[:, 3:7])
        # constructor: reset all, then first observation (hector_env.py:50-51)
        self.reset_idx(np.arange(n), init_pack)
        self.init_done = True
        self.compute_observations(init_pack)

    # ---- simulator <-> glue views
    def _refresh(self, full):
        s = self.state
        self.root = np.concatenate([s.root_pos, s.root_quat, s.root_linvel, s.root_angvel], 1).astype(F)
        self.dof_pos = s.q.astype(F)
        self.dof_vel = s.qd.astype(F)
        if full:
            self.rigid_state = self.phys.body_states(s).astype(F)
            self.contact_forces = self.phys.contact_force.astype(F)

    def _push_root(self, ids):
        s = self.state
        r = self.root.astype(s.q.dtype)
        s.root_pos[ids], s.root_quat[ids] = r[ids, 0:3], r[ids, 3:7]
        s.root_linvel[ids], s.root_angvel[ids] = r[ids, 7:10], r[ids, 10:13]

    def _push_dofs(self, ids):
        self.state.q[ids] = self.dof_pos[ids].astype(self.state.q.dtype)
        self.state.qd[ids] = self.dof_vel[ids].astype(self.state.q.dtype)

    # ---- gait clock (hector_env.py:70-88)
    def _phase(self):
        return (self.episode_length_buf.astype(F) * F(self.dt) / F(0.64)).astype(F)

    def _stance_mask(self):
        sin_pos = np.sin(F(2 * np.pi) * self._phase()).astype(F)
        m = np.zeros((self.n, 2), F)
        m[:, 0] = sin_pos >= 0
        m[:, 1] = sin_pos < 0
        m[np.abs(sin_pos) < 0.1] = 1
        return m

    # ---- step (hector_env.py:158-169 -> legged_robot.py:84-108)
    def step(self, actions, pack):
        RP, T = self.task.rp, self.task
        a = np.clip(np.asarray(actions, F), -F(T.clip), F(T.clip))
        delay = pack[RP["delay"]][:, None] * F(self.opts["action_delay"])
        a = (F(1) - delay) * a + delay * self.actions
        a = a + F(self.opts["action_noise"]) * pack[RP["act_noise"]:RP["act_noise"] + T.ndof].T * a
        self.actions = np.clip(a, -F(T.clip), F(T.clip)).astype(F)
        target = (self.actions * F(0.25) + self.task.default_q).astype(F)
        for _ in range(10):
            self.phys.substep(self.state, target.astype(np.float64), T.kp.astype(np.float64),
                              T.kd.astype(np.float64), T.torque_limit.astype(np.float64))
        # torque the reference reports = _compute_torques at the start of the last substep
        self.torques = self.phys.tau.astype(F)
        self._refresh(full=True)
        self.post_physics_step(pack)
        obs = np.clip(self.obs_buf, -F(T.clip), F(T.clip))
        priv = np.clip(self.priv_buf, -F(T.clip), F(T.clip))
        return obs, priv, self.rew_buf.copy(), self.reset_buf.copy()

    def post_physics_step(self, pack):
        n = self.n
        self.episode_length_buf += 1
        self.common_step_counter += 1
        q = self.root[:, 3:7]
        self.base_lin_vel = quat_rotate_inverse(q, self.root[:, 7:10])
        self.base_ang_vel = quat_rotate_inverse(q, self.root[:, 10:13])
        self.projected_gravity = quat_rotate_inverse(q, np.tile(np.array([[0, 0, -1]], F), (n, 1)))
        self.base_euler = euler_xyz_wrapped(q)
        # callback (legged_robot.py:303-319)
        ids = np.nonzero(self.episode_length_buf % 800 == 0)[0]
        self._resample_commands(ids, pack, "cmd_a")
        fwd = quat_apply(q, np.tile(np.array([[1, 0, 0]], F), (n, 1)))
        heading = np.arctan2(fwd[:, 1], fwd[:, 0]).astype(F)
        if self.opts["heading_command"]:                    # legged_robot.py:310-313
            self.commands[:, 2] = np.clip(F(0.5) * wrap_to_pi(self.commands[:, 3] - heading), -1, 1)
        if self.opts["push_robots"] and self.common_step_counter % 400 == 0:
            self._push_robots(pack)
        # termination (legged_robot.py:155-160)
        fn = np.sqrt(np.sum(self.contact_forces[:, self.task.term] ** 2, -1))
        self.reset_buf = np.any(fn > 1.0, 1)
        self.time_out_buf = self.episode_length_buf > self.max_episode_length
        self.reset_buf |= self.time_out_buf
        self.compute_reward()
        ids = np.nonzero(self.reset_buf)[0]
        self.reset_idx(ids, pack)
        self.compute_observations(pack)
        self.last_last_actions = self.last_actions.copy()
        self.last_actions = self.actions.copy()
        self.last_dof_vel = self.dof_vel.copy()
        self.last_root_vel = self.root[:, 7:13].copy()

    def _resample_commands(self, ids, pack, field):
        if len(ids) == 0:
            return
        o = self.task.rp[field]
        cr = self.opts["cmd_ranges"]
        self.commands[ids, 0] = unif(cr["lin_vel_x"][0], cr["lin_vel_x"][1], pack[o][ids])
        self.commands[ids, 1] = unif(cr["lin_vel_y"][0], cr["lin_vel_y"][1], pack[o + 1][ids])
        if self.opts["heading_command"]:                    # legged_robot.py:329-332
            self.commands[ids, 3] = unif(cr["heading"][0], cr["heading"][1], pack[o + 2][ids])
        else:
            self.commands[ids, 2] = unif(cr["ang_vel_yaw"][0], cr["ang_vel_yaw"][1], pack[o + 2][ids])
        nrm = np.sqrt(np.sum(self.commands[ids, :2] ** 2, 1))
        self.commands[ids, :2] *= (nrm > 0.2)[:, None]

    def _push_robots(self, pack):
        o = self.task.rp["push"]
        mx, ma = self.opts["max_push_vel_xy"], self.opts["max_push_ang_vel"]
        self.rand_push_force[:, :2] = unif(-mx, mx, pack[o:o + 2].T)
        self.root[:, 7:9] = self.rand_push_force[:, :2]
        self.rand_push_torque = unif(-ma, ma, pack[o + 2:o + 5].T)
        self.root[:, 10:13] = self.rand_push_torque
        self._push_root(np.arange(self.n))

    def _update_terrain_curriculum(self, ids, pack):
        """legged_robot.py:399-419.  randint_like(levels, max) arrives as a uniform: level = floor(u * max)."""
        if not self.init_done:
            return
        d = self.root[ids, :2] - self.env_origins[ids, :2]
        distance = np.sqrt(np.sum(d * d, 1, dtype=F), dtype=F)
        move_up = distance > F(self.env_length / 2)
        cn = np.sqrt(np.sum(self.commands[ids, :2] ** 2, 1, dtype=F), dtype=F)
        move_down = (distance < cn * F(self.max_episode_length * self.dt) * F(0.5)) & ~move_up
        lv = self.terrain_levels[ids] + move_up.astype(np.int64) - move_down.astype(np.int64)
        rnd = np.minimum(np.floor(pack[self.task.rp["level"]][ids] * F(self.max_terrain_level)).astype(np.int64), self.max_terrain_level - 1)
        self.terrain_levels[ids] = np.where(lv >= self.max_terrain_level, rnd, np.clip(lv, 0, None))
        self.env_origins[ids] = self.terrain_origins[self.terrain_levels[ids], self.terrain_types[ids]]

    # ---- reset (legged_robot.py:162-214, hector_env.py:256-261)
    def reset_idx(self, ids, pack):
        if len(ids) == 0:
            return
        if self.curriculum is not None:
            self._update_terrain_curriculum(ids, pack)
        o = self.task.rp["reset_q"]
        self.dof_pos[ids] = self.task.default_q + unif(-0.15, 0.15, pack[o:o + self.task.ndof].T[ids])
        self.dof_vel[ids] = 0
        self._push_dofs(ids)
        base_init = np.array([0, 0, self.task.base_init_z, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0], F)
        self.root[ids] = base_init
        self.root[ids, :3] += self.env_origins[ids]
        if self.custom_origins:
            o = self.task.rp["reset_xy"]
            self.root[ids, :2] += unif(-1.0, 1.0, pack[o:o + 2].T[ids])
        self._push_root(ids)
        self._resample_commands(ids, pack, "cmd_b")
        self.last_last_actions[ids] = 0
        self.actions[ids] = 0
        self.last_actions[ids] = 0
        self.last_dof_vel[ids] = 0
        self.feet_air_time[ids] = 0
        self.episode_length_buf[ids] = 0
        self.reset_buf[ids] = True
        self.extras_episode = {}
        for k in self.reward_order:
            self.extras_episode["rew_" + k] = np.mean(self.episode_sums[k][ids]) / F(24.0)
            self.episode_sums[k][ids] = 0
        self.time_outs_visible = self.time_out_buf.copy()
        self.base_euler = euler_xyz_wrapped(self.root[:, 3:7])
        self.projected_gravity[ids] = quat_rotate_inverse(self.root[ids, 3:7], np.tile(np.array([[0, 0, -1]], F), (len(ids), 1)))
        self.obs_hist[:, ids] = 0
        self.priv_hist[:, ids] = 0

    # ---- rewards (hector_env.py:277-539), alphabetical evaluation order (helpers.py:47 dir())
    def compute_reward(self):
        self.rew_buf = np.zeros(self.n, F)
        for name in self.reward_order:
            scale = F(self.reward_scale[name] * self.dt)
            rew = (getattr(self, "_reward_" + name)().astype(F) * scale).astype(F)
            self.rew_buf = (self.rew_buf + rew).astype(F)
            self.episode_sums[name] = (self.episode_sums[name] + rew).astype(F)
        if self.opts["only_positive_rewards"]:              # legged_robot.py:226-227
            self.rew_buf = np.maximum(self.rew_buf, F(0))

    def _contact(self):
        return self.contact_forces[:, self.task.feet, 2] > 5.0

    def _reward_action_smoothness(self):
        t1 = np.sum((self.last_actions - self.actions) ** 2, 1)
        t2 = np.sum((self.actions + self.last_last_actions - F(2) * self.last_actions) ** 2, 1)
        t3 = F(0.05) * np.sum(np.abs(self.actions), 1)
        return t1 + t2 + t3

    def _reward_base_acc(self):
        d = self.last_root_vel - self.root[:, 7:13]
        return np.exp(-np.sqrt(np.sum(d * d, 1)) * F(3))

    def _reward_base_height(self):
        sm = self._stance_mask()
        mh = np.sum(self.rigid_state[:, self.task.feet, 2] * sm, 1) / np.sum(sm, 1)
        bh = self.root[:, 2] - (mh - F(0.05))
        return np.exp(-np.abs(bh - F(self.task.base_height_target)) * F(100))

    def _reward_collision(self):
        fn = np.sqrt(np.sum(self.contact_forces[:, self.task.penal] ** 2, -1))
        return np.sum(F(1.0) * (fn > 0.1), 1)

    def _reward_default_joint_pos(self):
        jd = self.dof_pos - self.task.default_q
        nrm = lambda ab: np.sqrt(np.sum(jd[:, ab[0]:ab[1]] ** 2, 1))
        yr = nrm(self.task.djp_pairs[0]) + nrm(self.task.djp_pairs[1])             # hip yaw / roll of both legs
        yr = np.clip(yr - F(0.1), 0, 50)
        r = np.exp(-yr * F(100)) - F(0.01) * np.sqrt(np.sum(jd * jd, 1))
        if self.task.djp_arm_pairs:                                                # hector_w_arm_env.py:371-378
            ar = nrm(self.task.djp_arm_pairs[0]) + nrm(self.task.djp_arm_pairs[1])
            r = r + np.exp(-np.clip(ar - F(0.1), 0, 25) * F(2))
        return r

    def _reward_dof_acc(self):
        return np.sum(((self.last_dof_vel - self.dof_vel) / F(self.dt)) ** 2, 1)

    def _reward_dof_vel(self):
        return np.sum(self.dof_vel ** 2, 1)

    def _reward_feet_air_time(self):
        contact = self._contact()
        sm = self._stance_mask()
        filt = contact | (sm > 0) | self.last_contacts
        self.last_contacts = contact
        first = (self.feet_air_time > 0) * filt
        self.feet_air_time = (self.feet_air_time + F(self.dt)).astype(F)
        air = np.clip(self.feet_air_time, 0, 0.5) * first
        self.feet_air_time = (self.feet_air_time * ~filt).astype(F)
        return np.sum(air, 1)

    def _reward_feet_clearance(self):
        contact = self._contact()
        feet_z = self.rigid_state[:, self.task.feet, 2] - F(0.05)
        dz = feet_z - self.last_feet_z
        self.feet_height = (self.feet_height + dz).astype(F)
        self.last_feet_z = feet_z
        swing = F(1) - self._stance_mask()
        pos = np.abs(self.feet_height - F(0.06)) < 0.01
        r = np.sum(pos * swing, 1)
        self.feet_height = (self.feet_height * ~contact).astype(F)
        return r

    def _reward_feet_contact_forces(self):
        fn = np.sqrt(np.sum(self.contact_forces[:, self.task.feet] ** 2, -1))
        return np.sum(np.clip(fn - F(self.task.max_contact_force), 0, 400), 1)

    def _reward_feet_contact_number(self):
        contact = self._contact()
        sm = self._stance_mask()
        return np.mean(np.where(contact == sm, F(1), F(-0.3)), 1)

    def _dist_reward(self, idx, max_df):
        pos = self.rigid_state[:, idx, :2]
        d = np.sqrt(np.sum((pos[:, 0] - pos[:, 1]) ** 2, 1))
        dmin = np.clip(d - F(self.task.min_dist), -0.5, 0.0)
        dmax = np.clip(d - F(max_df), 0, 0.5)
        return (np.exp(-np.abs(dmin) * F(100)) + np.exp(-np.abs(dmax) * F(100))) / F(2)

    def _reward_feet_distance(self):
        return self._dist_reward(self.task.feet, 0.5)

    def _reward_foot_slip(self):
        contact = self._contact()
        sp = np.sqrt(np.sum(self.rigid_state[:, self.task.feet, 7:9] ** 2, 2))
        return np.sum(np.sqrt(sp) * contact, 1)

    def _reward_joint_pos(self):
        """hector_env.py:264-275; ref_dof_pos is what the LAST compute_observations left (compute_ref_state :90-111)."""
        d = (self.dof_pos - self.ref_dof_pos).astype(F)
        nn = np.sqrt(np.sum(d * d, 1, dtype=F), dtype=F)
        return np.exp(-F(2) * nn) - F(0.2) * np.clip(nn, 0, 0.5)

    def _reward_low_speed(self):
        """hector_env.py:468-499"""
        vx, cx = self.base_lin_vel[:, 0], self.commands[:, 0]
        a_s, a_c = np.abs(vx), np.abs(cx)
        low, high = a_s < F(0.5) * a_c, a_s > F(1.2) * a_c
        r = np.zeros(self.n, F)
        r[low] = -1.0
        r[high] = 0.0
        r[~(low | high)] = 1.2
        r[np.sign(vx) != np.sign(cx)] = -2.0
        return r * (a_c > 0.1)

    def _reward_track_vel_hard(self):
        """hector_env.py:407-424"""
        d = self.commands[:, :2] - self.base_lin_vel[:, :2]
        le = np.sqrt(np.sum(d * d, 1, dtype=F), dtype=F)
        ae = np.abs(self.commands[:, 2] - self.base_ang_vel[:, 2])
        return (np.exp(-le * F(10)) + np.exp(-ae * F(10))) / F(2) - F(0.2) * (le + ae)

    def _reward_vel_mismatch_exp(self):
        """hector_env.py:395-405"""
        lin = np.exp(-np.square(self.base_lin_vel[:, 2]) * F(10))
        w = self.base_ang_vel[:, :2]
        ang = np.exp(-np.sqrt(np.sum(w * w, 1, dtype=F), dtype=F) * F(5))
        return (lin + ang) / F(2)

    def _reward_knee_distance(self):
        return self._dist_reward(self.task.knees, 0.25)

    def _reward_orientation(self):
        a = np.exp(-np.sum(np.abs(self.base_euler[:, :2]), 1) * F(10))
        b = np.exp(-np.sqrt(np.sum(self.projected_gravity[:, :2] ** 2, 1)) * F(20))
        return (a + b) / F(2)

    def _reward_torques(self):
        return np.sum(self.torques ** 2, 1)

    def _reward_tracking_ang_vel(self):
        e = (self.commands[:, 2] - self.base_ang_vel[:, 2]) ** 2
        return np.exp(-e * F(5))

    def _reward_tracking_lin_vel(self):
        e = np.sum((self.commands[:, :2] - self.base_lin_vel[:, :2]) ** 2, 1)
        return np.exp(-e * F(5))

    # ---- observations (hector_env.py:172-254)
    def compute_observations(self, pack):
        ph = self._phase()
        # compute_ref_state (hector_env.py:90-111): the gait reference pose that _reward_joint_pos reads one step later
        sp = np.sin(F(2 * np.pi) * ph).astype(F)
        sl, sr = np.minimum(sp, 0), np.maximum(sp, 0)
        ref = np.zeros((self.n, self.task.ndof), F)
        s1 = F(0.17)                                        # rewards.target_joint_pos_scale (hector_config.py:151)
        ref[:, 2], ref[:, 3], ref[:, 4] = sl * s1, sl * (2 * s1), sl * s1
        rr = self.task.ref_right
        ref[:, rr], ref[:, rr + 1], ref[:, rr + 2] = sr * s1, sr * (2 * s1), sr * s1
        ref[np.abs(sp) < 0.1] = 0
        self.ref_dof_pos = ref
        sin_pos = np.sin(F(2 * np.pi) * ph).astype(F)[:, None]
        cos_pos = np.cos(F(2 * np.pi) * ph).astype(F)[:, None]
        sm = self._stance_mask()
        cm = self._contact().astype(F)
        cmd = np.concatenate([sin_pos, cos_pos, self.commands[:, :3] * np.array([2, 2, 1], F)], 1)
        qd = (self.dof_pos - self.task.default_q).astype(F)
        dq = self.dof_vel * F(0.05)
        if self.task.xbot_priv:                              # humanoid_env.py:218-236
            priv = np.concatenate([cmd, qd, dq, self.actions, (self.dof_pos - ref).astype(F), self.base_lin_vel * F(2), self.base_ang_vel,
                                   self.base_euler, self.rand_push_force[:, :2], self.rand_push_torque, self.env_frictions,
                                   self.body_mass / F(30.0), sm, cm], 1).astype(F)
        else:
            priv = np.concatenate([cmd, qd, dq, self.actions, self.base_lin_vel * F(2), self.base_ang_vel,
                                   self.base_euler, self.rigid_state[:, self.task.feet, :3].reshape(self.n, 6),
                                   self.rigid_state[:, self.task.feet, 7:10].reshape(self.n, 6), self.root[:, :3],
                                   self.rand_push_force[:, :2], self.rand_push_torque, self.env_frictions,
                                   self.body_mass / F(30.0), sm, cm], 1).astype(F)
        obs = np.concatenate([cmd, qd, dq, self.actions, self.base_ang_vel, self.base_euler], 1).astype(F)
        if self.add_noise:
            o = self.task.rp["obs_noise"]
            obs = (obs + pack[o:o + self.task.nobs].T * self.task.noise_vec * F(0.6)).astype(F)
        self.obs_hist = np.concatenate([self.obs_hist[1:], obs[None]], 0)
        self.priv_hist = np.concatenate([self.priv_hist[1:], priv[None]], 0)
        self.obs_buf = self.obs_hist.transpose(1, 0, 2).reshape(self.n, 15 * self.task.nobs)
        self.priv_buf = self.priv_hist.transpose(1, 0, 2).reshape(self.n, self.task.priv_stack * self.task.npriv)
