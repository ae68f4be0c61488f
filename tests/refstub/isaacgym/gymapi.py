"""Fake `gymapi`: one simulated scene whose rigid-body step is oracle/physics.py (float64).

Only what reference base_task.py / legged_robot.py / hector_env.py call is present.  The four state
tensors are float32 torch tensors on the CPU, exactly the views the reference wraps
(reference legged_robot.py:437-456); `simulate` advances the oracle by one 1 ms substep.
"""
import types

import numpy as np
import torch

from oracle import physics as _phys

SIM_PHYSX = 1
SIM_FLEX = 0
KEY_ESCAPE = 0
KEY_V = 1
DOF_MODE_EFFORT = 3


class Vec3:
    def __init__(self, x=0.0, y=0.0, z=0.0):
        self.x, self.y, self.z = float(x), float(y), float(z)


class Transform:
    def __init__(self, p=None, r=None):
        self.p = p or Vec3()
        self.r = r


class _Bag:
    pass


class SimParams(_Bag):
    def __init__(self):
        self.dt = 1 / 60.
        self.substeps = 1
        self.use_gpu_pipeline = False
        self.physx = _Bag()
        self.physx.use_gpu = False
        self.physx.num_subscenes = 0
        self.physx.num_threads = 0


class PlaneParams(_Bag):
    def __init__(self):
        self.normal = Vec3(0, 0, 1)
        self.static_friction = 1.0
        self.dynamic_friction = 1.0
        self.restitution = 0.0


class AssetOptions(_Bag):
    pass


class CameraProperties(_Bag):
    pass


class HeightFieldParams(_Bag):
    def __init__(self):
        self.transform = Transform()


TriangleMeshParams = HeightFieldParams


class _ShapeProps:
    def __init__(self):
        self.friction = 1.0


class _BodyProps:
    def __init__(self, mass):
        self.mass = mass


class Gym:
    def __init__(self):
        self.model = _phys.load_model()
        self.env = None           # back-reference set by the fixture generator
        self.n = 0
        self.shape_friction = []
        self.base_mass = []
        self.start_pos = []
        self.phys = None
        self.state = None
        self.torque_checks = 0
        self.substep_hook = None
        self.terrain_hf = None

    # ---- sim / asset creation
    def create_sim(self, *a):
        return "sim"

    def add_ground(self, sim, params):
        self.plane = params

    def add_heightfield(self, sim, heightsamples, params):
        """reference legged_robot.py:553-569"""
        from oracle.terrain import HeightField
        assert params.column_scale == params.row_scale and params.transform.p.x == params.transform.p.y
        self.terrain_hf = HeightField(np.asarray(heightsamples), params.column_scale, params.vertical_scale,
                                      -params.transform.p.x)

    def add_triangle_mesh(self, sim, vertices, triangles, params):
        """reference legged_robot.py:571-585: flattened float32 vertices / uint32 triangles of
        convert_heightfield_to_trimesh.  The grid is recovered from the vertex list (row 0 lies in the flat
        border, so its x is exactly 0 and the first x > 0 marks the row length)."""
        from oracle.terrain import HeightField
        v = np.asarray(vertices, np.float64).reshape(-1, 3)
        assert v.shape[0] == params.nb_vertices and np.asarray(triangles).size == 3 * params.nb_triangles
        cols = int(np.argmax(v[:, 0] > 1e-6))
        rows = v.shape[0] // cols
        assert rows * cols == v.shape[0] and 2 * (rows - 1) * (cols - 1) == params.nb_triangles
        hs = round(float(v[cols, 0] - v[0, 0]), 6)          # float32 vertex spacing -> the configured scale
        assert params.transform.p.x == params.transform.p.y
        # the reference hands over the slope-threshold mesh (utils/terrain.py:70-73): vertices next to a cliff are moved,
        # which the height function restates as vertical walls (oracle/terrain.py HeightField.contact); the threshold is
        # not recoverable from the mesh, the fixture generator sets it from the config it runs
        self.terrain_hf = HeightField(v[:, 2].reshape(rows, cols), hs, 1.0, -params.transform.p.x,
                                      wall_height=float(getattr(Gym, "trimesh_wall_height", 0.0)))

    def load_asset(self, sim, root, file, options):
        # the compiled model of the asset the task names: robot.urdf (hector) or robot_w_arm.urdf (hector_full)
        if "w_arm" in str(file):
            self.model = _phys.load_model(_phys.MODEL_FULL_JSON)
        elif "XBot" in str(file):
            self.model = _phys.load_model(_phys.MODEL_XBOT_JSON)
        self.nd, self.nbod = len(self.model["bodies"]) - 1, len(self.model["bodies"])
        return "asset"

    def get_asset_dof_count(self, asset):
        return self.nd

    def get_asset_rigid_body_count(self, asset):
        return self.nbod

    def get_asset_dof_properties(self, asset):
        B = self.model["bodies"][1:]
        props = np.zeros(self.nd, dtype=[("lower", "f4"), ("upper", "f4"), ("velocity", "f4"), ("effort", "f4")])
        for i, b in enumerate(B):
            props[i] = (b["lower"], b["upper"], b["velocity"], b["effort"])
        return props

    def get_asset_rigid_shape_properties(self, asset):
        return [_ShapeProps() for _ in self.model["contacts"]]

    def get_asset_rigid_body_names(self, asset):
        return [b["name"] for b in self.model["bodies"]]

    def get_asset_dof_names(self, asset):
        return [b["joint"] for b in self.model["bodies"][1:]]

    def create_env(self, sim, lo, hi, per_row):
        self.n += 1
        return self.n - 1

    def set_asset_rigid_shape_properties(self, asset, props):
        f = props[0].friction
        self._pending_friction = float(f.item() if hasattr(f, "item") else f)

    def create_actor(self, env, asset, pose, name, group, filt, seg):
        self.self_collision_filter = filt          # asset.self_collisions (legged_robot.py:656): 0 = links collide with each other
        self.shape_friction.append(getattr(self, "_pending_friction", 1.0))
        self.start_pos.append([pose.p.x, pose.p.y, pose.p.z])
        self.base_mass.append(self.model["bodies"][0]["mass"])
        return 0

    def set_actor_dof_properties(self, env, actor, props):
        pass

    def get_actor_rigid_body_properties(self, env, actor):
        return [_BodyProps(b["mass"]) for b in self.model["bodies"]]

    def set_actor_rigid_body_properties(self, env, actor, props, recomputeInertia=True):
        self.base_mass[env] = float(props[0].mass)

    def find_actor_rigid_body_handle(self, env, actor, name):
        return [b["name"] for b in self.model["bodies"]].index(name)

    def create_camera_sensor(self, env, props):
        return 0

    def prepare_sim(self, sim):
        n = self.n
        m0 = self.model["bodies"][0]["mass"]
        self.phys = _phys.HectorPhysics(n, base_mass_added=np.array(self.base_mass) - m0,
                                        shape_friction=np.array(self.shape_friction), terrain=self.terrain_hf, model=self.model)
        if getattr(self, "self_collision_filter", 1) == 0 and self.nd == 12:      # only the XBot-L model carries self-collision pairs
            self.phys.enable_self_collision()
        self.state = _phys.State(n, ndof=self.nd)
        self.state.root_pos[:] = np.array(self.start_pos)
        self.root_t = torch.zeros(n, 13)
        self.dof_t = torch.zeros(n * self.nd, 2)
        self.contact_t = torch.zeros(n * self.nbod, 3)
        self.body_t = torch.zeros(n * self.nbod, 13)
        self._publish()

    # ---- tensor API
    def acquire_actor_root_state_tensor(self, sim):
        return self.root_t

    def acquire_dof_state_tensor(self, sim):
        return self.dof_t

    def acquire_net_contact_force_tensor(self, sim):
        return self.contact_t

    def acquire_rigid_body_state_tensor(self, sim):
        return self.body_t

    def _publish(self):
        s = self.state
        self.root_t[:, 0:3] = torch.from_numpy(s.root_pos).float()
        self.root_t[:, 3:7] = torch.from_numpy(s.root_quat).float()
        self.root_t[:, 7:10] = torch.from_numpy(s.root_linvel).float()
        self.root_t[:, 10:13] = torch.from_numpy(s.root_angvel).float()
        d = self.dof_t.view(self.n, self.nd, 2)
        d[..., 0] = torch.from_numpy(s.q).float()
        d[..., 1] = torch.from_numpy(s.qd).float()
        self.body_t.view(self.n, self.nbod, 13)[:] = torch.from_numpy(self.phys.body_states(s)).float()
        self.contact_t.view(self.n, self.nbod, 3)[:] = torch.from_numpy(self.phys.contact_force).float()

    def refresh_dof_state_tensor(self, sim):
        pass   # tensors are republished at the end of every simulate / set call

    refresh_actor_root_state_tensor = refresh_net_contact_force_tensor = refresh_rigid_body_state_tensor = refresh_dof_state_tensor

    def set_dof_actuation_force_tensor(self, sim, t):
        self.tau_in = t.detach().clone().view(self.n, self.nd)

    def simulate(self, sim):
        env = self.env
        assert env is not None, "fixture generator must set gym.env"
        act = env.actions.double().numpy()
        target = act * env.cfg.control.action_scale + env.default_dof_pos.double().numpy()
        kp = env.p_gains[0].double().numpy()
        kd = env.d_gains[0].double().numpy()
        lim = env.torque_limits.double().numpy()
        if self.substep_hook is not None:
            self.substep_hook(self, target)
        self.phys.substep(self.state, target, kp, kd, lim)
        # the torque the reference handed over must be the one the physics applied (fp32 vs fp64 PD)
        assert np.allclose(self.tau_in.numpy(), self.phys.tau, rtol=1e-4, atol=2e-3), \
            np.abs(self.tau_in.numpy() - self.phys.tau).max()
        self.torque_checks += 1
        self._publish()

    def fetch_results(self, sim, wait):
        pass

    def set_dof_state_tensor_indexed(self, sim, t, ids, n):
        ids = ids.long().numpy()
        d = self.dof_t.view(self.n, self.nd, 2)
        self.state.q[ids] = d[ids, :, 0].double().numpy()
        self.state.qd[ids] = d[ids, :, 1].double().numpy()
        self._publish_bodies()

    def set_actor_root_state_tensor_indexed(self, sim, t, ids, n):
        ids = ids.long().numpy()
        r = self.root_t.double().numpy()
        self.state.root_pos[ids] = r[ids, 0:3]
        self.state.root_quat[ids] = r[ids, 3:7]
        self.state.root_linvel[ids] = r[ids, 7:10]
        self.state.root_angvel[ids] = r[ids, 10:13]
        self._publish_bodies()

    def set_actor_root_state_tensor(self, sim, t):
        self.set_actor_root_state_tensor_indexed(sim, t, torch.arange(self.n), self.n)

    def _publish_bodies(self):
        # PhysX does not refresh rigid_body_state / contact tensors on a set_* call: the reference
        # reads stale values there.  Keep them stale on purpose.
        pass


_GYM = None


def acquire_gym():
    global _GYM
    _GYM = Gym()
    return _GYM
