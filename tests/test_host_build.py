"""CPU: the HOST build of the simulator's single-source device code (oracle/host/hx_host.cpp: the text of
isaac_amd/csrc/hx_math.h, hx_dyn.h, hx_env.h compiled with g++) against the numpy oracle and the reference-generated
fixtures -- the kernels' own arithmetic checked without a GPU -- and the same text under AddressSanitizer + UBSan
("host build of single-source kernels under -fsanitize=address,undefined").
The host build is test infrastructure (cpu_baseline, sanitizers); the product never loads it."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.step_errors import check_step_errors

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _creation(fx):
    c = dict(friction=fx["init_shape_friction"], mass=fx["init_base_mass"], origins=fx["init_env_origins"], start=fx["init_start_pos"])
    if "terrain_heights" in fx:
        hs, vs, border = (float(x) for x in fx["terrain_params"])
        c["terrain"] = dict(heights=fx["terrain_heights"], horizontal_scale=hs, vertical_scale=vs, border_size=border)
    return c


def _env_from_fixture(fx, asan=False):
    from isaac_amd.envs.configs import HectorCfg, HectorFullCfg, XBotLCfg
    from oracle.host import HostEnv
    n, _, seed, sc0, noise = (int(x) for x in fx["meta"])
    task = str(fx["task"]) if "task" in fx else "hector"
    cfg = {"hector": HectorCfg, "hector_full": HectorFullCfg, "humanoid_ppo": XBotLCfg}[task]()
    cfg.env.num_envs = n
    cfg.noise.add_noise = bool(noise)
    cfg.terrain.mesh_type = "trimesh" if "terrain_heights" in fx else "plane"
    return HostEnv(cfg, creation=_creation(fx), init_pack=fx["packs"][0], task=task, asan=asan), n, sc0


@pytest.mark.parametrize("name,steps", [("env_rollout_a", 60), ("env_rollout_b", 40), ("env_rollout_c", 50), ("env_rollout_g", 80), ("env_rollout_h", 80)])
def test_host_build_replays_reference_fixture(name, steps):
    """Teacher-forced like tests/test_gpu_sim.py: every step starts from the fixture's recorded physics state."""
    fx = np.load(os.path.join(GOLD, name + ".npz"))
    env, n, sc0 = _env_from_fixture(fx)
    np.testing.assert_allclose(env.obs_buf, fx["init_obs_full"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(env.privileged_obs_buf, fx["init_priv_full"], rtol=0, atol=2e-5)
    env.episode_length_buf = fx["ep_len_init"].astype(np.int32)
    env.set_step_counter(sc0)
    F1, F2 = env.obs_f, env.priv_f
    errs = dict(obs=[], priv=[], rew=[], tau=[], contact=[])      # per (step, robot)
    for t in range(steps):
        if t > 0:
            env.set_state(fx["root"][t - 1], fx["q"][t - 1], fx["qd"][t - 1])
        obs, priv, rew, reset = env.step(fx["actions"][t], fx["packs"][t + 1])
        errs["obs"].append(np.abs(obs[:, -F1:] - fx["obs41"][t]).max(axis=1))
        errs["priv"].append(np.abs(priv[:, -F2:] - fx["priv70"][t]).max(axis=1))
        errs["rew"].append(np.abs(rew - fx["rew"][t]))
        errs["tau"].append(np.abs(env.torques - fx["torques"][t]).max(axis=1))
        errs["contact"].append(np.abs(env.contact_forces - fx["contact"][t]).reshape(n, -1).max(axis=1))
        assert np.array_equal(reset, fx["reset"][t].astype(bool)), f"reset flags differ at step {t}"
        assert np.array_equal(env.time_out_buf, fx["timeout"][t].astype(bool))
        assert np.array_equal(env.time_outs_visible, fx["timeouts_visible"][t].astype(bool))
        np.testing.assert_array_equal(env.episode_length_buf, fx["ep_len"][t])
    check_step_errors(name + " host build", errs)
    env.close()


def test_self_collision_pairs_are_exercised_by_fixture_h():
    """humanoid_ppo keeps self-collision on (asset.self_collisions = 0, humanoid_config.py:66): fixture H holds robots whose knees /
    feet touch (the generator counted the active pair contacts), and the same replay with the pairs switched off must leave the
    tight tier at those steps -- i.e. the test above really checks the pair contact, not its absence."""
    from isaac_amd.envs.configs import XBotLCfg
    from oracle.host import HostEnv
    fx = np.load(os.path.join(GOLD, "env_rollout_h.npz"))
    assert int(fx["self_contact_count"]) > 100
    n, _, seed, sc0, noise = (int(x) for x in fx["meta"])
    worst = {}
    for flag in (0, 1):
        cfg = XBotLCfg(); cfg.env.num_envs = n; cfg.noise.add_noise = bool(noise); cfg.terrain.mesh_type = "plane"
        cfg.asset.self_collisions = flag
        env = HostEnv(cfg, creation=_creation(fx), init_pack=fx["packs"][0], task="humanoid_ppo")
        env.episode_length_buf = fx["ep_len_init"].astype(np.int32)
        env.set_step_counter(sc0)
        w = 0.0
        for t in range(80):
            if t > 0:
                env.set_state(fx["root"][t - 1], fx["q"][t - 1], fx["qd"][t - 1])
            obs, priv, rew, reset = env.step(fx["actions"][t], fx["packs"][t + 1])
            w = max(w, float(np.abs(env.contact_forces - fx["contact"][t]).max()))
        worst[flag] = w
        env.close()
    print("worst contact-force error vs fixture H: pairs on %.3g N, pairs off %.3g N" % (worst[0], worst[1]))
    assert worst[0] < 1.0 and worst[1] > 10.0, worst


def test_host_build_free_run_stays_close():
    fx = np.load(os.path.join(GOLD, "env_rollout_a.npz"))
    env, n, sc0 = _env_from_fixture(fx)
    per_robot = np.zeros(n)
    for t in range(20):
        obs, priv, rew, reset = env.step(fx["actions"][t], fx["packs"][t + 1])
        per_robot = np.maximum(per_robot, np.abs(obs[:, -41:] - fx["obs41"][t]).max(axis=1))
    # 64 robots under unit-variance random actions are chaotic: a contact point that crosses its activation threshold one
    # substep earlier in fp32 than in float64 separates two trajectories for good.  The typical robot stays at round-off.
    print("free run, 20 steps: median %.2e, 90 %% %.2e, worst %.2e" % (np.median(per_robot), np.quantile(per_robot, 0.9), per_robot.max()))
    assert np.median(per_robot) < 2e-3 and np.quantile(per_robot, 0.9) < 3e-2
    env.close()


def test_host_build_under_address_and_ub_sanitizers():
    """The same source compiled with -fsanitize=address,undefined replays a terrain fixture (window fetch, wall contact,
    resets) in a child process; any report makes the child exit non-zero."""
    from oracle.host import build
    so = build(asan=True)
    code = (
        "import sys, os, numpy as np\\n"
        f"sys.path.insert(0, {ROOT!r})\\n"
        "from tests.test_host_build import _env_from_fixture, GOLD\\n"
        "for name, steps in (('env_rollout_c', 12), ('env_rollout_g', 12)):\\n"
        "    fx = np.load(os.path.join(GOLD, name + '.npz'))\\n"
        "    env, n, sc0 = _env_from_fixture(fx, asan=True)\\n"
        "    env.episode_length_buf = fx['ep_len_init'].astype(np.int32)\\n"
        "    env.set_step_counter(sc0)\\n"
        "    for t in range(steps):\\n"
        "        obs, priv, rew, reset = env.step(fx['actions'][t], fx['packs'][t + 1])\\n"
        "    assert np.all(np.isfinite(obs)) and np.all(np.isfinite(priv))\\n"
        "    env.close()\\n"
        "print('sanitized replay ok')\\n")
    asan_rt = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    env = dict(os.environ, LD_PRELOAD=asan_rt, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-c", code.replace("\\n", "\n")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "sanitized replay ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]


@pytest.mark.parametrize("task", ["hector", "hector_full"])
def test_second_stage_of_the_bounding_test_changes_nothing(task):
    """hx_dyn.h shape_gap_points (round 4): a shape that the bounding-sphere test flags is tested again, exactly, before it is put on
    the contact loop's visit list.  It may only skip visits that would have found no contact.  The host build with and without it
    (-DHX_NO_GAP2), free-running 128 robots on the default tile map from a drop (falls, wall contacts, resets included), must agree
    bit for bit in everything the step returns."""
    from isaac_amd.envs.configs import HectorCfg, HectorFullCfg
    from isaac_amd.utils.helpers import set_seed
    from oracle.host import HostEnv
    n, steps = 128, 70
    outs = []
    for variant in (False, "nogap2"):
        cfg = (HectorCfg if task == "hector" else HectorFullCfg)(); cfg.env.num_envs = n; cfg.seed = set_seed(11)
        np.random.seed(11)
        env = HostEnv(cfg, task=task, asan=variant)
        nd = env.nd
        rng = np.random.default_rng(5)
        rec = []
        for t in range(steps):
            obs, priv, rew, reset = env.step((1.0 * rng.standard_normal((n, nd))).astype(np.float32))
            rec.append((obs, priv, rew, reset.copy(), env.contact_forces, env.torques))
        outs.append(rec)
        env.close()
    resets = sum(int(r[3].sum()) for r in outs[0])
    nl = outs[0][0][5].shape[1] // 2
    feet = [nl, 2 * nl]                                  # body index (1 + joint) of the two feet: the last body of each leg chain (hector) ...
    if task != "hector":
        feet = [5, 5 + nl]                               # ... the fifth of each side for hector_full (arms follow the leg)
    other = [b for b in range(outs[0][0][4].shape[1]) if b not in feet]
    nonfoot = sum(int((np.abs(r[4][:, other, :]).sum(axis=2) > 0).sum()) for r in outs[0])
    print(f"{task}: {resets} resets, {nonfoot} (step, robot, body) contacts of bodies other than the feet")
    assert nonfoot > 20 and (resets >= 1 or task != "hector"), (resets, nonfoot)        # robots did touch with more than their feet (and hector fell)
    for t, (a, b) in enumerate(zip(*outs)):
        for x, y, name in zip(a, b, ("obs", "priv", "rewards", "resets", "contact forces", "torques")):
            np.testing.assert_array_equal(x, y, err_msg=f"step {t}: {name}")
