"""CPU: invariants of the float64 physics oracle (parity unpinned against PhysX -- see oracle/physics.py).
These are the checks that stand in for golden vectors of the closed-source simulator: conservation laws in
free flight, force balance at rest, analytic free fall, joint-limit and friction behaviour."""
import numpy as np

from oracle.physics import DT, GRAVITY, HectorPhysics, State

Q0 = np.array([0, 0, .785, -1.578, .785] * 2)
KP = np.array([40, 40, 60, 120, 20] * 2, float)
KD = np.array([3, 3, 5, 4, 1] * 2, float)
TL = 0.85 * np.array([33.5, 33.5, 33.5, 67, 33.5] * 2)


def _free(n, seed=0):
    ph = HectorPhysics(n)
    ph.q_lo[:], ph.q_hi[:], ph.v_max[:] = -100, 100, 1e9
    rng = np.random.default_rng(seed)
    s = State(n)
    s.root_pos[:, 2] = 5.0
    s.q[:] = Q0 + rng.uniform(-.1, .1, (n, 10))
    s.qd[:] = rng.uniform(-2, 2, (n, 10))
    s.root_angvel[:] = rng.uniform(-1, 1, (n, 3))
    s.root_linvel[:] = rng.uniform(-1, 1, (n, 3))
    return ph, s


def test_total_mass_matches_urdf():
    assert abs(HectorPhysics(1).mass.sum() - 15.0058) < 1e-3


def test_energy_and_momentum_conserved_in_free_flight():
    ph, s = _free(3)
    z = np.zeros((3, 10))
    ke, pe = ph.energy(s)
    P0, L0 = ph.momentum(s)
    dt, steps = 1e-4, 200
    for _ in range(steps):
        ph.substep(s, z, z, z, z + 100.0, dt=dt)
    ke1, pe1 = ph.energy(s)
    P1, L1 = ph.momentum(s)
    assert np.all(np.abs(ke1 + pe1 - ke - pe) < 5e-3)                 # first-order integrator, E ~ 740 J
    mg = ph.mass.sum(0)[:, None] * np.array([0, 0, GRAVITY])
    np.testing.assert_allclose(P1 - P0, mg * dt * steps, atol=1e-4)
    np.testing.assert_allclose((L1 - L0)[:, 2], 0, atol=1e-4)         # gravity exerts no torque about z


def test_free_fall_acceleration():
    ph = HectorPhysics(1)
    s = State(1)
    s.root_pos[:, 2] = 3.0
    s.q[:] = Q0
    z = np.zeros((1, 10))
    for _ in range(100):
        ph.substep(s, np.broadcast_to(Q0, (1, 10)), KP, KD, TL)
    np.testing.assert_allclose(s.root_linvel[0, 2], GRAVITY * 0.1, rtol=1e-3)


def test_static_force_balance_equals_weight():
    n = 2
    ph = HectorPhysics(n, base_mass_added=[0.0, 3.0], shape_friction=[1.0, 0.3])
    s = State(n)
    s.root_pos[:, 2] = 0.56
    s.q[:] = Q0
    for _ in range(2500):                                           # topples (ankle Kp < m g h) and comes to rest
        ph.substep(s, np.broadcast_to(Q0, (n, 10)), KP, KD, TL)
    np.testing.assert_allclose(ph.contact_force[:, :, 2].sum(1), -GRAVITY * ph.mass.sum(0), rtol=2e-3)
    assert np.all(np.abs(s.root_linvel) < 5e-2)


def test_joint_limits_hold():
    ph = HectorPhysics(1)
    s = State(1)
    s.root_pos[:, 2] = 3.0
    s.q[:] = Q0
    tgt = np.broadcast_to(Q0 + 5.0, (1, 10))                        # drive every joint far past its upper limit
    for _ in range(400):
        ph.substep(s, tgt, KP, KD, TL)
    assert np.all(s.q[0] < ph.q_hi + 0.06)                          # soft limit: tau_max / k = 57/2000 rad overshoot


def test_friction_cone_limits_tangential_force():
    ph = HectorPhysics(1, shape_friction=[0.1])
    s = State(1)
    s.root_pos[:, 2] = 0.56
    s.q[:] = Q0
    for _ in range(300):
        ph.substep(s, np.broadcast_to(Q0, (1, 10)), KP, KD, TL)
    f = ph.contact_force[0, [5, 10]]
    mu = 0.5 * (0.6 + 0.1)
    assert np.all(np.linalg.norm(f[:, :2], axis=1) <= mu * np.abs(f[:, 2]) * 1.3 + 1.0)


def test_tree_generic_on_the_arm_model():
    """The oracle's dynamics is written for any tree: on the 18-DoF hector-with-arms model (SURVEY 8f-4 groundwork,
    tools/compile_urdf.py --full) free flight conserves energy and momentum, and the DoF order is the one the
    reference task indexes (hector_w_arm_env.py:371-373: L leg 0-4, L arm 5-8, R leg 9-13, R arm 14-17)."""
    from oracle.physics import MODEL_FULL_JSON, load_model
    model = load_model(MODEL_FULL_JSON)
    joints = [b["joint"] for b in model["bodies"][1:]]
    assert [j[0] for j in joints] == list("LLLLLLLLLRRRRRRRRR")
    assert ["shoulder" in j or "elbow" in j for j in joints] == ([False] * 5 + [True] * 4) * 2
    n = 2
    ph = HectorPhysics(n, model=model)
    assert ph.ndof == 18 and abs(ph.mass.sum(0)[0] - model["total_mass"]) < 1e-9
    ph.q_lo[:], ph.q_hi[:], ph.v_max[:] = -100, 100, 1e9
    rng = np.random.default_rng(1)
    s = State(n, ndof=18)
    s.root_pos[:, 2] = 5.0
    s.q[:] = rng.uniform(-.3, .3, (n, 18))
    s.qd[:] = rng.uniform(-2, 2, (n, 18))
    s.root_angvel[:] = rng.uniform(-1, 1, (n, 3))
    s.root_linvel[:] = rng.uniform(-1, 1, (n, 3))
    z = np.zeros((n, 18))
    ke, pe = ph.energy(s)
    P0, L0 = ph.momentum(s)
    dt, steps = 1e-4, 200
    for _ in range(steps):
        ph.substep(s, z, z, z, z + 100.0, dt=dt)
    ke1, pe1 = ph.energy(s)
    P1, L1 = ph.momentum(s)
    assert np.all(np.abs(ke1 + pe1 - ke - pe) < 5e-3)
    mg = ph.mass.sum(0)[:, None] * np.array([0, 0, GRAVITY])
    np.testing.assert_allclose(P1 - P0, mg * dt * steps, atol=1e-4)
    np.testing.assert_allclose((L1 - L0)[:, 2], 0, atol=1e-4)


def test_rotated_joint_frames_on_the_xbot_model():
    """SURVEY 8f-4 groundwork for `humanoid_ppo`: XBot-L's twelve revolute joints turn about the local z of frames that are
    rotated against their parents (and two about -z), which neither hector asset needs.  tools/compile_urdf.py --xbot
    records the constant rotation per joint ("rot"); the oracle applies it in the kinematics.  Checks: with the base at the
    reference config's standing height (XBotLCfg.rewards.base_height_target = 0.89) the zero pose has its soles on the ground, the legs are
    mirror images, and free flight conserves energy and momentum on this tree too."""
    from oracle.physics import MODEL_XBOT_JSON, load_model
    model = load_model(MODEL_XBOT_JSON)
    assert [b["joint"] for b in model["bodies"][1:]] == [f"{s}_{j}_joint" for s in ("left", "right") for j in
                                                         ("leg_roll", "leg_yaw", "leg_pitch", "knee", "ankle_pitch", "ankle_roll")]
    assert all("rot" in b for b in model["bodies"][1:])
    n = 2
    ph = HectorPhysics(n, model=model)
    assert ph.ndof == 12 and abs(ph.mass.sum(0)[0] - model["total_mass"]) < 1e-9
    s = State(n, ndof=12)
    s.root_pos[:, 2] = 0.89
    R, p, _, _ = ph.kinematics(s)
    world = {}
    for body, pts in ph.contacts:
        world.setdefault(int(body), []).append(p[body][0] + pts @ R[body][0].T)
    world = {b: np.concatenate(v) for b, v in world.items()}
    assert sorted(world) == [0, 3, 4, 6, 9, 10, 12]          # base (box + head + arms), thighs, calves, feet (XBot-L.urdf's enabled collisions)
    fl, fr = world[6], world[12]
    for b in (0, 3, 4, 9, 10):                                # nothing but the feet reaches the ground in the standing pose
        assert world[b][:, 2].min() > 0.03, b                 # the calf mesh ends 4.8 cm above the ground, at the ankle
    assert abs(fl[:, 2].min()) < 0.01 and abs(fr[:, 2].min()) < 0.01            # soles at the ground (measured: 5 mm)
    assert fl[:, 2].max() < 0.15 and fl[:, 1].mean() > 0.05 > -0.05 > fr[:, 1].mean()
    np.testing.assert_allclose(np.sort(fl[:, 0]), np.sort(fr[:, 0]), atol=2e-3)   # mirror images in the sagittal plane
    np.testing.assert_allclose(np.sort(fl[:, 1]), np.sort(-fr[:, 1]), atol=2e-3)
    for i in range(1, 7):                                                        # every left body mirrors its right twin
        np.testing.assert_allclose(p[i][0] * [1, -1, 1], p[i + 6][0], atol=2e-3)
    assert p[4][0][2] < p[3][0][2] < p[1][0][2]                                  # knee below hip pitch below hip roll
    # a pitch joint moves the foot in the sagittal plane, whichever way its frame is turned
    s2 = State(n, ndof=12)
    s2.root_pos[:, 2] = 0.89
    s2.q[:, 2] = 0.3
    _, p2, _, _ = ph.kinematics(s2)
    d = p2[6][0] - p[6][0]
    assert abs(d[1]) < 1e-3 and abs(d[0]) > 0.1
    # dynamics on the rotated tree
    ph.q_lo[:], ph.q_hi[:], ph.v_max[:] = -100, 100, 1e9
    rng = np.random.default_rng(2)
    s = State(n, ndof=12)
    s.root_pos[:, 2] = 5.0
    s.q[:] = rng.uniform(-.3, .3, (n, 12))
    s.qd[:] = rng.uniform(-2, 2, (n, 12))
    s.root_angvel[:] = rng.uniform(-1, 1, (n, 3))
    s.root_linvel[:] = rng.uniform(-1, 1, (n, 3))
    z = np.zeros((n, 12))
    ke, pe = ph.energy(s)
    P0, L0 = ph.momentum(s)
    dt, steps = 1e-4, 200
    for _ in range(steps):
        ph.substep(s, z, z, z, z + 1000.0, dt=dt)
    ke1, pe1 = ph.energy(s)
    P1, L1 = ph.momentum(s)
    assert np.all(np.abs(ke1 + pe1 - ke - pe) < 2e-2)                             # E ~ 2.6 kJ
    mg = ph.mass.sum(0)[:, None] * np.array([0, 0, GRAVITY])
    np.testing.assert_allclose(P1 - P0, mg * dt * steps, atol=5e-4)
    np.testing.assert_allclose((L1 - L0)[:, 2], 0, atol=5e-4)
