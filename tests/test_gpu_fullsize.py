"""GPU: checks at BASELINE.json's full sizes (4096 envs, T = 60, minibatches of 61 440 rows).

The oracle's env step is too slow for 4096 robots x many steps, so the simulator is checked through size-independent
properties (batch independence, determinism); the learner is small enough per sample that one full-size update is
compared with the numpy oracle directly (about a minute of host time)."""
import numpy as np
import pytest

from isaac_amd import capi
from isaac_amd.algo.ppo import PPO, ActorCritic
from isaac_amd.envs.configs import HectorCfg, HectorFullCfg
from isaac_amd.envs.hector_env import HectorFreeEnv, HectorFullFreeEnv, creation_randomisation
from isaac_amd.utils.helpers import set_seed

pytestmark = pytest.mark.gpu
N, T = 4096, 60


TASKS = {"hector": (HectorCfg, HectorFreeEnv), "hector_full": (HectorFullCfg, HectorFullFreeEnv)}


def _creation(seed, n, task="hector"):
    """Creation data of the default config (tile map) for n robots, drawn once on the host."""
    from isaac_amd.envs.terrain import HumanoidTerrain
    cfg_cls, env_cls = TASKS[task]
    cfg = cfg_cls()
    cfg.terrain.mesh_type = "trimesh"                    # HectorFullCfg's own default is the plane
    cfg.env.num_envs = n
    set_seed(seed)
    ter = HumanoidTerrain(cfg.terrain, n)
    probe = env_cls.__new__(env_cls)
    origins = probe._terrain_origins(cfg, n, ter)
    fr, ms, start = creation_randomisation(cfg, n, origins, env_cls.BASE_MASS)
    grid = dict(heights=ter.heightsamples, horizontal_scale=cfg.terrain.horizontal_scale,
                vertical_scale=cfg.terrain.vertical_scale, border_size=cfg.terrain.border_size)
    return dict(friction=fr, mass=ms, origins=origins, start=start, terrain=grid,
                terrain_levels=probe.terrain_levels, terrain_types=probe.terrain_types)


def _env(creation, lo, hi, seed, task="hector"):
    cfg_cls, env_cls = TASKS[task]
    cfg = cfg_cls()
    cfg.terrain.mesh_type = "trimesh"
    cfg.env.num_envs = len(creation["friction"])
    cfg.seed = seed
    return env_cls(cfg, creation=creation, env_range=(lo, hi))


def _roll(env, actions, ep_len):
    env.episode_length_buf = ep_len
    out = []
    for a in actions:
        o, p, r, d, _ = env.step(a)
        out.append((o.numpy().copy(), p.numpy().copy(), r.numpy().copy(), d.numpy().copy()))
    return out


@pytest.mark.parametrize("task", list(TASKS))
def test_full_batch_is_deterministic_and_batch_independent(hxlib, task):
    """4096 robots on the default tile map, 12 steps with resets and time-outs in them:
    (1) two runs from the same seed are bit-identical; (2) robots 1024..1087 simulated alone (a 64-robot simulator with
    env_id_offset 1024) produce bit-identical observations, rewards and resets -- nothing in a robot's step depends on
    the batch it is in (random streams are keyed by the global env id)."""
    seed = 5
    cr = _creation(seed, N, task)
    rng = np.random.default_rng(1)
    acts = (0.6 * rng.standard_normal((12, N, 10 if task == "hector" else 18))).astype(np.float32)
    ep = rng.integers(0, 2400, N).astype(np.int32)
    ep[::97] = 2398                                      # time-outs inside the window
    a = _env(cr, 0, N, seed, task); ra = _roll(a, acts, ep); a.close()
    b = _env(cr, 0, N, seed, task); rb = _roll(b, acts, ep); b.close()
    for (o1, p1, r1, d1), (o2, p2, r2, d2) in zip(ra, rb):
        assert np.array_equal(o1, o2) and np.array_equal(p1, p2) and np.array_equal(r1, r2) and np.array_equal(d1, d2)
    assert sum(int(d.sum()) for *_, d in ra) > 20        # the window does contain resets
    lo, hi = 1024, 1088
    c = _env(cr, lo, hi, seed, task); rc = _roll(c, acts[:, lo:hi], ep[lo:hi]); c.close()
    for t, ((o1, p1, r1, d1), (o3, p3, r3, d3)) in enumerate(zip(ra, rc)):
        # one documented coupling: extras["time_outs"] staleness is per simulator, it does not enter obs / reward / done
        assert np.array_equal(o1[lo:hi], o3), f"obs differ at step {t}"
        assert np.array_equal(p1[lo:hi], p3) and np.array_equal(r1[lo:hi], r3) and np.array_equal(d1[lo:hi], d3)


def test_full_size_update_matches_oracle(hxlib):
    """One PPO iteration at N = 4096, T = 60 (245 760 samples, 4 minibatches of 61 440, 1 epoch) against the numpy oracle:
    values / log-probs at rollout time, returns and normalised advantages, both mean losses, the learning-rate decisions
    and the parameters after the 4 Adam steps."""
    from oracle.ppo import ActorCriticOracle, PPOOracle
    seed = 11
    init = ActorCriticOracle.default_init(np.random.default_rng(seed))
    ac = ActorCritic(615, 1050, 10, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[768, 256, 128], init_noise_std=1.0)
    ac.load_state_dict(init.state_dict())
    kw = dict(num_learning_epochs=1, num_mini_batches=4, learning_rate=1e-4)
    alg = PPO(ac, clip_param=0.2, gamma=0.994, lam=0.9, value_loss_coef=1.0, entropy_coef=0.001, max_grad_norm=1.0,
              use_clipped_value_loss=True, schedule="adaptive", desired_kl=0.01, **kw)
    alg.init_storage(N, T, [615], [1050], [10])
    orc = PPOOracle(ActorCriticOracle.default_init(np.random.default_rng(seed)), N, T, **kw)
    rng = np.random.default_rng(2)
    base_o = rng.standard_normal((N, 615)).astype(np.float32)
    base_p = rng.standard_normal((N, 1050)).astype(np.float32)
    for t in range(T):
        o = (base_o + 0.3 * rng.standard_normal((N, 615))).astype(np.float32)
        p = (base_p + 0.3 * rng.standard_normal((N, 1050))).astype(np.float32)
        e = rng.standard_normal((N, 10)).astype(np.float32)
        a = alg.act(o, p, eps=e).numpy()
        a2 = orc.act(o, p, e)
        if t % 20 == 0:
            np.testing.assert_allclose(a, a2, rtol=0, atol=1e-4)
        r = rng.uniform(0, 0.05, N).astype(np.float32)
        d = rng.uniform(size=N) < 0.02
        to = d & (rng.uniform(size=N) < 0.3)
        alg.process_env_step(r, d.astype(np.uint8), {"time_outs": to.astype(np.uint8)})
        orc.process_env_step(r, d, to)
    alg.compute_returns(p)
    orc.compute_returns(p)
    adv = alg.buffer(6, (T, N)).numpy()
    np.testing.assert_allclose(adv, orc.advantages, rtol=1e-4, atol=3e-4)
    assert abs(float(adv.mean())) < 1e-4 and abs(float(adv.std(ddof=1)) - 1.0) < 1e-4          # size-independent property
    perm = np.random.default_rng(3).permutation(T * N).astype(np.int32)
    vl, sl = alg.update(perm=perm)
    vl2, sl2 = orc.update(perm)
    assert abs(vl - vl2) < 1e-4 * max(1.0, abs(vl2)) and abs(sl - sl2) < 1e-4, (vl, vl2, sl, sl2)
    assert abs(alg.learning_rate / orc.lr - 1) < 1e-6
    sd = ac.state_dict()
    for k, v in orc.ac.state_dict().items():
        d = np.abs(sd[k] - v)
        assert d.max() <= 4 * 2.0 * 1e-2 * 1e-2 + 8 * orc.lr, (k, d.max())      # at most lr-sized noise per Adam step
        assert np.mean(d > 5e-6) < 2e-3, (k, float(np.mean(d > 5e-6)))
    alg.close()


def test_bf16_mode_matches_emulating_oracle(hxlib):
    """BASELINE config 4 (hx_ppo_set_compute_dtype 1): N = 4096, T = 16, 4 minibatches of 16 384 rows.  The oracle rounds
    the operands of the same products to bf16 (oracle/ppo.py bf16=True).  What remains is fp32 summation order PLUS the
    occasional activation that sits on a bf16 rounding boundary and rounds the other way in the next layer (one such
    flip moves a value by ~1e-4): values 1e-3, advantages 5e-3, losses 2e-3 relative.  Against the fp32 oracle the
    values differ by O(1e-2), i.e. the mode really is bf16."""
    from oracle.ppo import ActorCriticOracle, PPOOracle
    seed, Tb = 12, 16
    init = ActorCriticOracle.default_init(np.random.default_rng(seed))
    ac = ActorCritic(615, 1050, 10, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[768, 256, 128], init_noise_std=1.0)
    ac.load_state_dict(init.state_dict())
    kw = dict(num_learning_epochs=1, num_mini_batches=4, learning_rate=1e-4)
    alg = PPO(ac, clip_param=0.2, gamma=0.994, lam=0.9, value_loss_coef=1.0, entropy_coef=0.001, max_grad_norm=1.0,
              use_clipped_value_loss=True, schedule="adaptive", desired_kl=0.01, mlp_dtype="bf16", **kw)
    alg.init_storage(N, Tb, [615], [1050], [10])
    orc = PPOOracle(ActorCriticOracle.default_init(np.random.default_rng(seed)), N, Tb, bf16=True, **kw)
    ref32 = ActorCriticOracle.default_init(np.random.default_rng(seed))
    rng = np.random.default_rng(4)
    for t in range(Tb):
        o = rng.standard_normal((N, 615)).astype(np.float32)
        p = rng.standard_normal((N, 1050)).astype(np.float32)
        e = rng.standard_normal((N, 10)).astype(np.float32)
        a = alg.act(o, p, eps=e).numpy()
        np.testing.assert_allclose(a, orc.act(o, p, e), rtol=0, atol=1e-3)          # rollout actor rounds like the update
        r = rng.uniform(0, 0.05, N).astype(np.float32)
        d = rng.uniform(size=N) < 0.02
        alg.process_env_step(r, d.astype(np.uint8), {"time_outs": np.zeros(N, np.uint8)})
        orc.process_env_step(r, d, np.zeros(N, bool))
    alg.compute_returns(p)
    orc.compute_returns(p)
    vals = alg.buffer(1, (Tb, N)).numpy()
    np.testing.assert_allclose(vals, orc.values, rtol=0, atol=1e-3)
    v32 = ref32.evaluate(p)[:, 0]
    assert 1e-4 < np.abs(vals[-1] - v32).max() < 5e-2                               # bf16, not fp32
    np.testing.assert_allclose(alg.buffer(6, (Tb, N)).numpy(), orc.advantages, rtol=0, atol=5e-3)
    perm = np.random.default_rng(3).permutation(Tb * N).astype(np.int32)
    vl, sl = alg.update(perm=perm)
    vl2, sl2 = orc.update(perm)
    assert abs(vl - vl2) < 2e-3 * max(1.0, abs(vl2)) and abs(sl - sl2) < 2e-3, (vl, vl2, sl, sl2)
    assert abs(alg.learning_rate / orc.lr - 1) < 1e-6
    sd = ac.state_dict()
    for k, v in orc.ac.state_dict().items():
        d = np.abs(sd[k] - v)
        assert np.mean(d > 5e-5) < 2e-2, (k, float(np.mean(d > 5e-5)), d.max())
    alg.close()


@pytest.mark.parametrize("n", [1, 33, 100])
def test_ragged_batch_sizes(hxlib, n):
    """Batch sizes that fill neither a wave (32 robots) nor a workgroup: the first n robots of a 128-robot batch,
    simulated alone, give bit-identical observations / rewards / resets (lanes past the batch must not disturb it)."""
    seed = 7
    cr = _creation(seed, 128)
    rng = np.random.default_rng(2)
    acts = (0.6 * rng.standard_normal((8, 128, 10))).astype(np.float32)
    ep = rng.integers(0, 2400, 128).astype(np.int32)
    ep[:4] = 2397
    full = _env(cr, 0, 128, seed)
    ra = _roll(full, acts, ep)
    full.close()
    part = _env(cr, 0, n, seed)
    rb = _roll(part, acts[:, :n], ep[:n])
    part.close()
    for (o1, p1, r1, d1), (o2, p2, r2, d2) in zip(ra, rb):
        assert np.array_equal(o1[:n], o2) and np.array_equal(p1[:n], p2) and np.array_equal(r1[:n], r2) and np.array_equal(d1[:n], d2)


def test_small_ragged_learner_runs(hxlib):
    """N = 33 robots, T = 8, 3 minibatches of 88 rows: every learner kernel with row counts that are not multiples of
    its tiles; compared with the oracle end to end."""
    from oracle.ppo import ActorCriticOracle, PPOOracle
    n, t = 33, 8
    init = ActorCriticOracle.default_init(np.random.default_rng(3))
    ac = ActorCritic(615, 1050, 10, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[768, 256, 128], init_noise_std=1.0)
    ac.load_state_dict(init.state_dict())
    kw = dict(num_learning_epochs=2, num_mini_batches=3, learning_rate=1e-4)
    alg = PPO(ac, gamma=0.994, lam=0.9, entropy_coef=0.001, schedule="adaptive", desired_kl=0.01, **kw)
    alg.init_storage(n, t, [615], [1050], [10])
    orc = PPOOracle(ActorCriticOracle.default_init(np.random.default_rng(3)), n, t, **kw)
    rng = np.random.default_rng(8)
    for _ in range(t):
        o, p, e = rng.standard_normal((n, 615)).astype(np.float32), rng.standard_normal((n, 1050)).astype(np.float32), rng.standard_normal((n, 10)).astype(np.float32)
        np.testing.assert_allclose(alg.act(o, p, eps=e).numpy(), orc.act(o, p, e), rtol=0, atol=5e-5)
        r, d = rng.uniform(0, 0.05, n).astype(np.float32), rng.uniform(size=n) < 0.1
        alg.process_env_step(r, d.astype(np.uint8), {})
        orc.process_env_step(r, d)
    alg.compute_returns(p)
    orc.compute_returns(p)
    np.testing.assert_allclose(alg.buffer(6, (t, n)).numpy(), orc.advantages, rtol=1e-4, atol=2e-4)
    perm = rng.permutation(n * t).astype(np.int32)
    vl, sl = alg.update(perm=perm)
    vl2, sl2 = orc.update(perm)
    assert abs(vl - vl2) < 1e-4 * max(1, abs(vl2)) and abs(sl - sl2) < 1e-4
    alg.close()
