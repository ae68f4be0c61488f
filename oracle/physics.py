"""ORACLE (test infrastructure, never shipped) -- rigid-body dynamics of the hector biped.

PARITY UNPINNED.  The reference delegates this step to the closed-source `isaacgym` binary
(`self.gym.simulate(self.sim)`, reference humanoid/envs/base/legged_robot.py:93-100; solver settings
humanoid/envs/custom/hector_config.py:103-120).  That package is absent from /root/reference and from
this image, it has no golden vectors, and its TGS articulation solver is not documented to the level
that would allow a bitwise restatement.  What this file pins instead is the *definition* of the
dynamics the HIP kernel implements (DESIGN.md "Physics model"):

  * 11-body floating-base articulation from the reference's URDF (tools/compile_urdf.py),
    gravity (0,0,-9.81), 1 ms semi-implicit Euler, joint velocity clamp at the URDF limits;
  * PD actuation torque exactly as reference legged_robot.py:339-355 (`_compute_torques`), with its
    q/qd derivatives treated linearly-implicitly while unclipped;
  * soft joint limits and soft ground contact (plane z=0) as linearly-implicit spring-dampers,
    regularised Coulomb friction with mu = mean(terrain mu, per-env shape mu) (PhysX "average" combine).

The algorithm here is deliberately NOT the one the kernel uses: the kernel runs Featherstone's
articulated-body algorithm in fp32, this oracle assembles the joint-space system with the composite
rigid body algorithm + recursive Newton-Euler in float64 and solves it densely.  Agreement of the two
to fp32 round-off checks the hardest code in the product against an independent derivation.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import json
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
MODEL_JSON = os.path.join(os.path.dirname(_HERE), "isaac_amd", "assets", "hector_model.json")
# hector with arms (task hector_full, reference hector_w_arm_config.py:29): 19 bodies / 18 DoF, DoF order L leg, L arm,
# R leg, R arm; no collision shapes compiled yet (tools/compile_urdf.py --full)
MODEL_FULL_JSON = os.path.join(os.path.dirname(_HERE), "isaac_amd", "assets", "hector_full_model.json")
MODEL_XBOT_JSON = MODEL_FULL_JSON.replace("hector_full_model.json", "xbot_model.json")   # XBot-L, 12 DoF (oracle only)

# ---- simulation constants (mirrored in isaac_amd/csrc/hx_sim.hip; DESIGN.md lists them) ----
GRAVITY = -9.81           # reference legged_robot_config.py:184
DT = 1e-3                 # reference hector_config.py:104
CONTACT_KN = 4.0e4        # N/m per contact point
CONTACT_DN = 4.0e2        # N s/m per contact point
FRICTION_VEPS = 2.0e-2    # m/s, width of the viscous zone of the regularised Coulomb law
LIMIT_K = 2.0e3           # Nm/rad soft joint limit
LIMIT_D = 2.0e1           # Nm s/rad
TERRAIN_MU = 0.6          # reference hector_config.py:50-51
# the contact inputs the reference states for PhysX (hector_config.py:113-117; include/hx_sim.h hx_sim_cfg): the spring part of a
# point's normal force is capped at the force that pushes it out at MAX_DEPEN_VEL; a point within CONTACT_OFFSET of the
# surface is a contact whose approach speed beyond gap / dt is damped; shapes rest at REST_OFFSET
MAX_DEPEN_VEL = 1.0       # max_depenetration_velocity
CONTACT_OFFSET = 0.01     # contact_offset
REST_OFFSET = 0.0         # rest_offset


def load_model(path=MODEL_JSON):
    with open(path) as f:
        return json.load(f)


def skew(v):
    z = np.zeros_like(v[..., 0])
    return np.stack([np.stack([z, -v[..., 2], v[..., 1]], -1),
                     np.stack([v[..., 2], z, -v[..., 0]], -1),
                     np.stack([-v[..., 1], v[..., 0], z], -1)], -2)


def axis_rot(k, q):
    """Rotation matrix R_k(q) (child body -> parent coordinates) for a revolute joint about axis k."""
    c, s = np.cos(q), np.sin(q)
    o, z = np.ones_like(q), np.zeros_like(q)
    if k == 0:
        rows = [[o, z, z], [z, c, -s], [z, s, c]]
    elif k == 1:
        rows = [[c, z, s], [z, o, z], [-s, z, c]]
    else:
        rows = [[c, -s, z], [s, c, z], [z, z, o]]
    return np.stack([np.stack(r, -1) for r in rows], -2)


def quat_to_mat(q):
    """xyzw quaternion -> rotation matrix (body -> world)."""
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    return np.stack([
        np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
        np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
        np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)], -2)


def mat_to_quat(R):
    """rotation matrix -> xyzw quaternion with w >= 0 (batched, branch-free by largest component)."""
    m00, m11, m22 = R[..., 0, 0], R[..., 1, 1], R[..., 2, 2]
    cand = np.stack([1 + m00 - m11 - m22, 1 - m00 + m11 - m22, 1 - m00 - m11 + m22, 1 + m00 + m11 + m22], -1)
    i = np.argmax(cand, -1)
    out = np.zeros(R.shape[:-2] + (4,), R.dtype)
    for k in range(4):
        sel = i == k
        if not np.any(sel):
            continue
        Rs = R[sel]
        t = np.sqrt(np.maximum(cand[sel, k], 1e-30)) * 2
        if k == 0:
            q = np.stack([t / 4, (Rs[:, 0, 1] + Rs[:, 1, 0]) / t, (Rs[:, 0, 2] + Rs[:, 2, 0]) / t, (Rs[:, 2, 1] - Rs[:, 1, 2]) / t], -1)
        elif k == 1:
            q = np.stack([(Rs[:, 0, 1] + Rs[:, 1, 0]) / t, t / 4, (Rs[:, 1, 2] + Rs[:, 2, 1]) / t, (Rs[:, 0, 2] - Rs[:, 2, 0]) / t], -1)
        elif k == 2:
            q = np.stack([(Rs[:, 0, 2] + Rs[:, 2, 0]) / t, (Rs[:, 1, 2] + Rs[:, 2, 1]) / t, t / 4, (Rs[:, 1, 0] - Rs[:, 0, 1]) / t], -1)
        else:
            q = np.stack([(Rs[:, 2, 1] - Rs[:, 1, 2]) / t, (Rs[:, 0, 2] - Rs[:, 2, 0]) / t, (Rs[:, 1, 0] - Rs[:, 0, 1]) / t, t / 4], -1)
        out[sel] = q
    out *= np.where(out[..., 3:4] < 0, -1.0, 1.0)
    return out


def crm(v):
    """Spatial cross-product operator for motion vectors, v x (6x6)."""
    w, l = skew(v[..., :3]), skew(v[..., 3:])
    z = np.zeros_like(w)
    return np.concatenate([np.concatenate([w, z], -1), np.concatenate([l, w], -1)], -2)


class State:
    """Per-env generalized state.  All arrays have leading dim N."""

    def __init__(self, n, dtype=np.float64, ndof=10):
        self.root_pos = np.zeros((n, 3), dtype)
        self.root_quat = np.zeros((n, 4), dtype)
        self.root_quat[:, 3] = 1
        self.root_linvel = np.zeros((n, 3), dtype)   # world frame, velocity of the base-link origin
        self.root_angvel = np.zeros((n, 3), dtype)   # world frame
        self.q = np.zeros((n, ndof), dtype)
        self.qd = np.zeros((n, ndof), dtype)

    def copy(self):
        s = State(self.q.shape[0], self.q.dtype, self.q.shape[1])
        for k, v in self.__dict__.items():
            setattr(s, k, v.copy())
        return s


class HectorPhysics:
    def __init__(self, n, base_mass_added=None, shape_friction=None, dtype=np.float64, model=None, terrain=None):
        self.n = n
        self.terrain = terrain      # None = ground plane z=0; else oracle.terrain.HeightField
        self.dtype = dtype
        self.model = model or load_model()
        B = self.model["bodies"]
        self.nb = len(B)
        self.parent = [b["parent"] for b in B]
        self.axis = [b.get("axis", -1) for b in B]
        self.offset = np.array([b.get("offset", [0, 0, 0]) for b in B], dtype)
        self.rot = [np.array(b["rot"], dtype) if "rot" in b else None for b in B]
        self.q_lo = np.array([b["lower"] for b in B[1:]], dtype)
        self.q_hi = np.array([b["upper"] for b in B[1:]], dtype)
        self.v_max = np.array([b["velocity"] for b in B[1:]], dtype)
        added = np.zeros(n, dtype) if base_mass_added is None else np.asarray(base_mass_added, dtype)
        self.shape_mu = np.full(n, 1.0, dtype) if shape_friction is None else np.asarray(shape_friction, dtype)
        # spatial inertia of each body about its own frame origin [N,6,6]
        self.I = np.zeros((self.nb, n, 6, 6), dtype)
        self.mass = np.zeros((self.nb, n), dtype)
        for i, b in enumerate(B):
            m0 = b["mass"]
            m = np.full(n, m0, dtype)
            Ic = np.broadcast_to(np.array(b["inertia_com"], dtype), (n, 3, 3)).copy()
            if i == 0:
                # randomised payload: reference legged_robot.py:295-301 adds to props[0].mass and asks
                # PhysX to recompute the inertia; here the tensor scales with the mass (same COM).
                m = m + added
                Ic = Ic * (m / m0)[:, None, None]
            c = np.broadcast_to(np.array(b["com"], dtype), (n, 3))
            cx = skew(c)
            self.mass[i] = m
            self.I[i, :, :3, :3] = Ic + m[:, None, None] * (cx @ np.swapaxes(cx, -1, -2))
            self.I[i, :, :3, 3:] = m[:, None, None] * cx
            self.I[i, :, 3:, :3] = m[:, None, None] * np.swapaxes(cx, -1, -2)
            self.I[i, :, 3:, 3:] = m[:, None, None] * np.eye(3, dtype=dtype)
        self.contacts = [(c["body"], np.array(c["points"], dtype)) for c in self.model["contacts"]]
        # self-collision sphere pairs (left body, right body, centre in the left / right body frame, radius): set by
        # enable_self_collision(); empty = the links do not collide with each other (hector_config.py:37)
        self.self_pairs = []
        # outputs of the last substep
        self.contact_force = np.zeros((n, self.nb, 3), dtype)
        self.ndof = self.nb - 1
        self.tau = np.zeros((n, self.ndof), dtype)

    # ------------------------------------------------------------------ kinematics
    def enable_self_collision(self):
        """asset.self_collisions = 0 (humanoid_config.py:66).  The product's proxy for the link pairs of a biped that can touch
        (isaac_amd/csrc/hx_dyn.h ModelXBot): knee against knee -- 6 cm spheres on the knee joints -- and foot against foot -- 5 cm
        spheres on the centre of the ankle_roll hull's bounding box.  Only the XBot-L model carries pairs."""
        names = [b["name"] for b in self.model["bodies"]]
        nl = (self.nb - 1) // 2
        by_body = {}
        for c in self.model["contacts"]:
            by_body.setdefault(c["body"], []).extend(c["points"])

        def bbox_centre(body):
            pts = np.array(by_body[body], self.dtype)
            return (pts.min(0) + pts.max(0)) / 2
        self.self_pairs = []
        self.self_contact_count = 0           # active (robot, pair side, substep) triples so far: fixtures assert that it is not zero
        for local, at_shape, radius in ((3, False, 0.06), (5, True, 0.05)):
            bl, br = 1 + local, 1 + nl + local
            cl = bbox_centre(bl) if at_shape else np.zeros(3, self.dtype)
            cr = bbox_centre(br) if at_shape else np.zeros(3, self.dtype)
            self.self_pairs.append((bl, br, cl, cr, radius))

    def kinematics(self, s):
        """Returns per body: R (body->world) [nb,N,3,3], p (world) [nb,N,3], Xup (parent->body 6x6),
        v (spatial velocity, body coords) [nb,N,6]."""
        n, nb = self.n, self.nb
        R = np.zeros((nb, n, 3, 3), self.dtype)
        p = np.zeros((nb, n, 3), self.dtype)
        Xup = np.zeros((nb, n, 6, 6), self.dtype)
        v = np.zeros((nb, n, 6), self.dtype)
        R[0] = quat_to_mat(s.root_quat)
        p[0] = s.root_pos
        R0T = np.swapaxes(R[0], -1, -2)
        v[0, :, :3] = np.einsum("nij,nj->ni", R0T, s.root_angvel)
        v[0, :, 3:] = np.einsum("nij,nj->ni", R0T, s.root_linvel)
        for i in range(1, nb):
            lam, k = self.parent[i], self.axis[i]
            Rj = axis_rot(k, s.q[:, i - 1])          # child -> parent
            if self.rot[i] is not None:              # joint frame rotated against the parent's (XBot-L; never for hector)
                Rj = self.rot[i] @ Rj
            E = np.swapaxes(Rj, -1, -2)             # parent -> child coordinates
            r = np.broadcast_to(self.offset[i], (n, 3))
            Xup[i, :, :3, :3] = E
            Xup[i, :, 3:, 3:] = E
            Xup[i, :, 3:, :3] = -E @ skew(r)
            R[i] = R[lam] @ Rj
            p[i] = p[lam] + np.einsum("nij,j->ni", R[lam], self.offset[i])
            v[i] = np.einsum("nij,nj->ni", Xup[i], v[lam])
            v[i, :, k] += s.qd[:, i - 1]
        return R, p, Xup, v

    def body_states(self, s):
        """rigid_body_state-like view: pos, quat(xyzw), linvel, angvel per body in the world frame
        (what `acquire_rigid_body_state_tensor` exposes, reference legged_robot.py:440,456)."""
        R, p, _, v = self.kinematics(s)
        out = np.zeros((self.n, self.nb, 13), self.dtype)
        for i in range(self.nb):
            out[:, i, 0:3] = p[i]
            out[:, i, 3:7] = mat_to_quat(R[i])
            out[:, i, 7:10] = np.einsum("nij,nj->ni", R[i], v[i, :, 3:])
            out[:, i, 10:13] = np.einsum("nij,nj->ni", R[i], v[i, :, :3])
        out[:, 0, 3:7] = s.root_quat
        return out

    # ------------------------------------------------------------------ one 1 ms substep
    def pd_torque(self, s, target, kp, kd, tau_lim):
        """reference legged_robot.py:339-355; `target` = action*action_scale + default_dof_pos."""
        raw = kp * (target - s.q) - kd * s.qd
        tau = np.clip(raw, -tau_lim, tau_lim)
        unclipped = (raw == tau)
        return tau, unclipped

    def substep(self, s, target, kp, kd, tau_lim, dt=DT):
        n, nb = self.n, self.nb
        dtp = self.dtype
        R, p, Xup, v = self.kinematics(s)
        S = np.zeros((nb, 6), dtp)
        for i in range(1, nb):
            S[i, self.axis[i]] = 1.0

        # ---- actuation + soft joint limits (joint space, linearly implicit)
        tau, unclipped = self.pd_torque(s, target, kp, kd, tau_lim)
        self.tau = tau.copy()
        beta = np.where(unclipped, dt * (kd + dt * kp), 0.0) * np.ones((n, self.ndof), dtp)
        c_lim = LIMIT_D + LIMIT_K * dt
        lo_pen = self.q_lo - s.q
        hi_pen = s.q - self.q_hi
        t_lo = LIMIT_K * lo_pen - c_lim * s.qd
        act_lo = (lo_pen > 0) & (t_lo > 0)
        t_hi = -LIMIT_K * hi_pen - c_lim * s.qd
        act_hi = (hi_pen > 0) & (t_hi < 0)
        tau_j = tau + np.where(act_lo, t_lo, 0.0) + np.where(act_hi, t_hi, 0.0)
        beta = beta + np.where(act_lo | act_hi, c_lim * dt, 0.0)

        # ---- contacts: explicit part f0 (spatial force on the body, body coords) and implicit 6x6 B
        f0 = np.zeros((nb, n, 6), dtp)
        Bm = np.zeros((nb, n, 6, 6), dtp)
        mu = 0.5 * (TERRAIN_MU + self.shape_mu)
        c_n = CONTACT_DN + CONTACT_KN * dt
        point_rec = []
        for body, pts in self.contacts:
            Rb = R[body]
            for r in pts:
                pw = p[body] + np.einsum("nij,j->ni", Rb, r)
                if self.terrain is None:
                    nrm_b = Rb[:, 2, :]                  # world z axis in body coords = R^T e_z
                    pen = -pw[:, 2]
                else:
                    # distance to the plane of the terrain triangle under the point, along its normal
                    pen, nrm_w = self.terrain.contact(pw[:, 0], pw[:, 1], pw[:, 2])
                    nrm_b = np.einsum("nji,nj->ni", Rb, nrm_w).astype(dtp)
                    pen = pen.astype(dtp)
                vb = v[body][:, 3:] + np.cross(v[body][:, :3], r)     # point velocity, body coords
                vn = np.einsum("ni,ni->n", vb, nrm_b)
                pen = pen + REST_OFFSET
                spring = CONTACT_KN * np.maximum(pen, 0.0)
                if MAX_DEPEN_VEL > 0:
                    spring = np.minimum(spring, c_n * MAX_DEPEN_VEL)
                fn0 = spring - c_n * (vn + np.maximum(-pen, 0.0) / dt)
                act = (pen > -CONTACT_OFFSET) & (fn0 > 0)
                vt = vb - vn[:, None] * nrm_b
                vt_norm = np.sqrt(np.einsum("ni,ni->n", vt, vt))
                c_t = mu * fn0 / np.maximum(vt_norm, FRICTION_VEPS)
                fvec = fn0[:, None] * nrm_b - c_t[:, None] * vt
                fvec = np.where(act[:, None], fvec, 0.0)
                nn = nrm_b[:, :, None] * nrm_b[:, None, :]
                K = dt * (c_n * nn + c_t[:, None, None] * (np.eye(3, dtype=dtp) - nn))
                K = np.where(act[:, None, None], K, 0.0)
                Xc = np.concatenate([np.broadcast_to(-skew(r), (n, 3, 3)),
                                     np.broadcast_to(np.eye(3, dtype=dtp), (n, 3, 3))], -1)   # [N,3,6]
                XcT = np.swapaxes(Xc, -1, -2)
                f0[body] += np.einsum("nij,nj->ni", XcT, fvec)
                Bm[body] += XcT @ K @ Xc
                point_rec.append((body, Xc, K, fvec, Rb))

        # ---- self-collision sphere pairs: the ground contact's normal law without friction; each body sees the other one's
        # velocity explicitly and its own linearly-implicitly
        for bl, br, cl, cr, radius in self.self_pairs:
            cw = {b: p[b] + np.einsum("nij,j->ni", R[b], c) for b, c in ((bl, cl), (br, cr))}
            vw = {b: np.einsum("nij,nj->ni", R[b], v[b][:, 3:] + np.cross(v[b][:, :3], c)) for b, c in ((bl, cl), (br, cr))}
            for own, oth, rc in ((bl, br, cl), (br, bl, cr)):
                d = cw[own] - cw[oth]
                dist = np.sqrt(np.einsum("ni,ni->n", d, d))
                nrm = d / np.maximum(dist, 1e-6)[:, None]
                pen = 2.0 * radius - dist + REST_OFFSET
                vn = np.einsum("ni,ni->n", vw[own] - vw[oth], nrm)
                spring = CONTACT_KN * np.maximum(pen, 0.0)
                if MAX_DEPEN_VEL > 0:
                    spring = np.minimum(spring, c_n * MAX_DEPEN_VEL)
                fn0 = spring - c_n * (vn + np.maximum(-pen, 0.0) / dt)
                act = (pen > -CONTACT_OFFSET) & (fn0 > 0)
                self.self_contact_count += int(act.sum())
                nrm_b = np.einsum("nji,nj->ni", R[own], nrm)
                fvec = np.where(act[:, None], fn0[:, None] * nrm_b, 0.0)
                K = np.where(act[:, None, None], dt * c_n * nrm_b[:, :, None] * nrm_b[:, None, :], 0.0)
                Xc = np.concatenate([np.broadcast_to(-skew(rc), (n, 3, 3)), np.broadcast_to(np.eye(3, dtype=dtp), (n, 3, 3))], -1)
                XcT = np.swapaxes(Xc, -1, -2)
                f0[own] += np.einsum("nij,nj->ni", XcT, fvec)
                Bm[own] += XcT @ K @ Xc
                point_rec.append((own, Xc, K, fvec, R[own]))

        # ---- bias forces: RNEA with zero generalized acceleration, gravity as base acceleration
        g_b = np.zeros((n, 6), dtp)
        g_b[:, 3:] = R[0][:, 2, :] * GRAVITY                        # R0^T (0,0,g)
        a_vp = np.zeros((nb, n, 6), dtp)                            # velocity-product accelerations
        a_rn = np.zeros((nb, n, 6), dtp)
        a_rn[0] = -g_b
        f = np.zeros((nb, n, 6), dtp)
        for i in range(nb):
            if i > 0:
                lam = self.parent[i]
                ci = np.einsum("nij,j->ni", crm(v[i]), S[i]) * s.qd[:, i - 1:i]
                a_vp[i] = np.einsum("nij,nj->ni", Xup[i], a_vp[lam]) + ci
                a_rn[i] = np.einsum("nij,nj->ni", Xup[i], a_rn[lam]) + ci
            Iv = np.einsum("nij,nj->ni", self.I[i], v[i])
            f[i] = np.einsum("nij,nj->ni", self.I[i], a_rn[i]) - np.einsum("nji,nj->ni", crm(v[i]), Iv)
            # contact: explicit force, and the implicit part acting on the velocity-product + gravity
            # share of the body's true spatial acceleration (a_true = J nu_dot + a_vp)
            f[i] -= f0[i] - np.einsum("nij,nj->ni", Bm[i], a_vp[i])
        C = np.zeros((n, 6 + self.ndof), dtp)
        for i in range(nb - 1, 0, -1):
            C[:, 6 + i - 1] = f[i][:, self.axis[i]]
            f[self.parent[i]] += np.einsum("nji,nj->ni", Xup[i], f[i])
        C[:, :6] = f[0]

        # ---- joint-space inertia: CRBA on (I + B)
        Ic = self.I + Bm
        Ic = Ic.copy()
        for i in range(nb - 1, 0, -1):
            XT = np.swapaxes(Xup[i], -1, -2)
            Ic[self.parent[i]] += XT @ Ic[i] @ Xup[i]
        H = np.zeros((n, 6 + self.ndof, 6 + self.ndof), dtp)
        H[:, :6, :6] = Ic[0]
        for i in range(1, nb):
            F = Ic[i][:, :, self.axis[i]]
            H[:, 5 + i, 5 + i] = F[:, self.axis[i]]
            j = i
            while self.parent[j] > 0:
                F = np.einsum("nji,nj->ni", Xup[j], F)
                j = self.parent[j]
                H[:, 5 + i, 5 + j] = H[:, 5 + j, 5 + i] = F[:, self.axis[j]]
            F = np.einsum("nji,nj->ni", Xup[j], F)
            H[:, :6, 5 + i] = F
            H[:, 5 + i, :6] = F
        idx = np.arange(self.ndof)
        H[:, 6 + idx, 6 + idx] += beta

        rhs = -C
        rhs[:, 6:] += tau_j
        nud = np.linalg.solve(H, rhs[..., None])[..., 0]
        a0 = nud[:, :6]            # true base spatial acceleration (gravity entered through a_rn[0])
        qdd = nud[:, 6:]

        # ---- contact forces actually applied (for net_contact_force): f = f0 - K Xc a_true
        a_true = np.zeros((nb, n, 6), dtp)
        a_true[0] = a0
        for i in range(1, nb):
            lam = self.parent[i]
            ci = np.einsum("nij,j->ni", crm(v[i]), S[i]) * s.qd[:, i - 1:i]
            a_true[i] = np.einsum("nij,nj->ni", Xup[i], a_true[lam]) + ci
            a_true[i][:, self.axis[i]] += qdd[:, i - 1]
        self.contact_force[:] = 0
        for body, Xc, K, fvec, Rb in point_rec:
            fb = fvec - np.einsum("nij,nj->ni", K @ Xc, a_true[body])
            self.contact_force[:, body] += np.einsum("nij,nj->ni", Rb, fb)

        # ---- integrate (semi-implicit Euler)
        wb, vb0 = v[0][:, :3], v[0][:, 3:]
        acc_ang = a_true[0][:, :3]
        acc_lin = a_true[0][:, 3:] + np.cross(wb, vb0)
        s.root_angvel = s.root_angvel + dt * np.einsum("nij,nj->ni", R[0], acc_ang)
        s.root_linvel = s.root_linvel + dt * np.einsum("nij,nj->ni", R[0], acc_lin)
        s.qd = np.clip(s.qd + dt * qdd, -self.v_max, self.v_max)
        s.q = s.q + dt * s.qd
        s.root_pos = s.root_pos + dt * s.root_linvel
        w = s.root_angvel
        x, y, z, ww = s.root_quat[:, 0], s.root_quat[:, 1], s.root_quat[:, 2], s.root_quat[:, 3]
        # q_dot = 0.5 * (w,0) (x) q  for a world-frame angular velocity
        dq = 0.5 * np.stack([w[:, 0] * ww + w[:, 1] * z - w[:, 2] * y,
                             w[:, 1] * ww + w[:, 2] * x - w[:, 0] * z,
                             w[:, 2] * ww + w[:, 0] * y - w[:, 1] * x,
                             -(w[:, 0] * x + w[:, 1] * y + w[:, 2] * z)], -1)
        qn = s.root_quat + dt * dq
        s.root_quat = qn / np.sqrt(np.einsum("ni,ni->n", qn, qn))[:, None]
        return s

    # ------------------------------------------------------------------ diagnostics
    def energy(self, s):
        """Kinetic + gravitational potential energy (no contact/limit potential)."""
        R, p, _, v = self.kinematics(s)
        ke = 0.0
        pe = 0.0
        for i in range(self.nb):
            ke = ke + 0.5 * np.einsum("ni,nij,nj->n", v[i], self.I[i], v[i])
            com = np.array(self.model["bodies"][i]["com"], self.dtype)
            zc = p[i][:, 2] + np.einsum("nj,j->n", R[i][:, 2, :], com)
            pe = pe - self.mass[i] * GRAVITY * zc
        return ke, pe

    def momentum(self, s):
        """World-frame linear momentum and angular momentum about the world origin."""
        R, p, _, v = self.kinematics(s)
        P = np.zeros((self.n, 3), self.dtype)
        Lw = np.zeros((self.n, 3), self.dtype)
        for i in range(self.nb):
            h = np.einsum("nij,nj->ni", self.I[i], v[i])       # [n; f] about body origin, body coords
            hl = np.einsum("nij,nj->ni", R[i], h[:, 3:])
            ha = np.einsum("nij,nj->ni", R[i], h[:, :3])
            P += hl
            Lw += ha + np.cross(p[i], hl)
        return P, Lw
