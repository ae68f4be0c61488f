"""GPU: the HIP learner on the network shapes of the reference's sibling tasks, against the numpy oracle.

  hector_full   obs 65x15=975, privileged 94x15=1410, 18 actions, actor [768,512,128], critic [768,768,768]
                (humanoid/envs/custom/hector_w_arm_config.py:10-17, 213-214)
  humanoid_ppo  obs 47x15=705, privileged 73x3=219, 12 actions, actor [512,256,128], critic [768,256,128]
                (humanoid/envs/custom/humanoid_config.py:40-46, 236-237)
  odd           small widths that hit every padding rule (inputs not multiples of 4, unequal last hidden widths)

The learner itself is shape-generic; only the env-step kernel is hector-specific.  These cases exercise: last hidden
widths that differ between actor and critic (loss head, rollout heads, slab layout), more than 16 actions, the
non-fused rollout actor, and input widths that need padding.  Tolerances as in tests/test_gpu_ppo.py.
"""
import numpy as np
import pytest

from isaac_amd.algo.ppo import PPO, ActorCritic
from oracle.ppo import ActorCriticOracle, PPOOracle
from tests.ppo_inputs import rollout_inputs

pytestmark = pytest.mark.gpu

SHAPES = {
    "hector_full": (975, 1410, 18, (768, 512, 128), (768, 768, 768)),
    "humanoid_ppo": (705, 219, 12, (512, 256, 128), (768, 256, 128)),
    "odd": (50, 31, 3, (64, 72, 64), (128, 36, 192)),
}


def _pair(name, seed, T, N, epochs, nmb, lr):
    no, npv, na, ah, ch = SHAPES[name]
    mk = lambda: ActorCriticOracle.default_init(np.random.default_rng(seed), no, npv, na, ah, ch, 1.0)
    init = mk()
    ac = ActorCritic(no, npv, na, actor_hidden_dims=list(ah), critic_hidden_dims=list(ch), init_noise_std=1.0)
    ac.load_state_dict(init.state_dict())
    alg = PPO(ac, num_learning_epochs=epochs, num_mini_batches=nmb, clip_param=0.2, gamma=0.994, lam=0.9, value_loss_coef=1.0,
              entropy_coef=0.001, learning_rate=lr, max_grad_norm=1.0, use_clipped_value_loss=True, schedule="adaptive",
              desired_kl=0.01)
    alg.init_storage(N, T, [no], [npv], [na])
    orc = PPOOracle(mk(), N, T, num_learning_epochs=epochs, num_mini_batches=nmb, learning_rate=lr)
    return ac, alg, orc, rollout_inputs(seed, T, N, no, npv, na)


@pytest.mark.parametrize("name", list(SHAPES))
def test_rollout_and_update_match_the_oracle(hxlib, name):
    seed, T, N, epochs, nmb = 5, 6, 48, 2, 3
    na = SHAPES[name][2]
    ac, alg, orc, inp = _pair(name, seed, T, N, epochs, nmb, 1e-4)
    sd = ac.state_dict()
    for k, v in orc.ac.state_dict().items():          # round trip through the padded device layout
        np.testing.assert_array_equal(sd[k], v)
    for t in range(T):
        a = alg.act(inp["obs"][t], inp["priv"][t], eps=inp["eps"][t]).numpy()
        ao = orc.act(inp["obs"][t], inp["priv"][t], inp["eps"][t])
        np.testing.assert_allclose(a, ao, rtol=0, atol=5e-5)
        alg.process_env_step(inp["rewards"][t], inp["dones"][t].astype(np.uint8), {"time_outs": inp["time_outs"][t].astype(np.uint8)})
        orc.process_env_step(inp["rewards"][t], inp["dones"][t], inp["time_outs"][t])
    alg.compute_returns(inp["priv"][T])
    orc.compute_returns(inp["priv"][T])
    np.testing.assert_allclose(alg.buffer(1, (T, N)).numpy(), orc.values, rtol=0, atol=5e-5)
    np.testing.assert_allclose(alg.buffer(2, (T, N)).numpy(), orc.logp, rtol=0, atol=1e-4)
    np.testing.assert_allclose(alg.buffer(3, (T, N, na)).numpy(), orc.mu, rtol=0, atol=5e-5)
    np.testing.assert_allclose(alg.buffer(5, (T, N)).numpy(), orc.returns, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(alg.buffer(6, (T, N)).numpy(), orc.advantages, rtol=1e-4, atol=2e-4)
    perm = np.random.default_rng(9).permutation(T * N).astype(np.int32)
    mvl, msl = alg.update(perm=perm)
    ovl, osl = orc.update(perm)
    assert abs(mvl - ovl) <= 1e-4 * max(1.0, abs(ovl))
    assert abs(msl - osl) <= 1e-4
    assert abs(alg.learning_rate / orc.lr - 1.0) < 1e-6
    sd = ac.state_dict()
    steps = epochs * nmb
    for k, v in orc.ac.state_dict().items():
        d = np.abs(sd[k] - v)
        assert d.max() <= 2 * steps * orc.lr, (k, d.max())            # Adam's normalised step on round-off gradients
        assert np.mean(d > 3e-6) < 2e-3, (k, float(np.mean(d > 3e-6)))


def test_inference_head_matches_on_sibling_shape(hxlib):
    ac, alg, orc, inp = _pair("hector_full", 2, 2, 40, 1, 1, 1e-4)
    mu = ac.act_inference(inp["obs"][0]).numpy()
    _, mu_o, _ = orc.ac.act(inp["obs"][0], np.zeros_like(inp["eps"][0]))
    np.testing.assert_allclose(mu, mu_o, rtol=0, atol=5e-5)


def test_more_than_32_actions_is_refused(hxlib):
    ac = ActorCritic(40, 40, 33, actor_hidden_dims=[64, 64, 64], critic_hidden_dims=[64, 64, 64], init_noise_std=1.0)
    alg = PPO(ac, num_learning_epochs=1, num_mini_batches=1)
    with pytest.raises(RuntimeError, match="num_actions"):
        alg.init_storage(8, 2, [40], [40], [33])


def test_failed_create_leaves_the_library_usable(hxlib):
    """A configuration error found after allocation has begun (row stride smaller than the padded input width) frees the
    half-built learner; the next learner works."""
    ac = ActorCritic(40, 40, 4, actor_hidden_dims=[64, 64, 64], critic_hidden_dims=[64, 64, 64], init_noise_std=1.0)
    alg = PPO(ac, num_learning_epochs=1, num_mini_batches=1)
    with pytest.raises(RuntimeError, match="obs_ld"):
        alg.init_storage(8, 2, [40], [40], [4], obs_ld=12)
    ac, alg, orc, inp = _pair("odd", 1, 2, 8, 1, 1, 1e-4)
    a = alg.act(inp["obs"][0], inp["priv"][0], eps=inp["eps"][0]).numpy()
    np.testing.assert_allclose(a, orc.act(inp["obs"][0], inp["priv"][0], inp["eps"][0]), rtol=0, atol=5e-5)


def test_option_branches_match_the_oracle(hxlib):
    """Branches the hector defaults never take: unclipped value loss (ppo.py:157-158), fixed learning-rate schedule
    (ppo.py:136 skipped), a gradient-norm limit small enough that clip_grad_norm_ always scales (ppo.py:173), other
    loss coefficients and discount factors."""
    no, npv, na, ah, ch = SHAPES["odd"]
    seed, T, N, epochs, nmb, lr = 17, 5, 40, 2, 2, 3e-4
    kw = dict(num_learning_epochs=epochs, num_mini_batches=nmb, clip_param=0.1, gamma=0.97, lam=0.8, value_loss_coef=0.5,
              entropy_coef=0.01, learning_rate=lr, max_grad_norm=0.05, use_clipped_value_loss=False, schedule="fixed", desired_kl=0.01)
    mk = lambda: ActorCriticOracle.default_init(np.random.default_rng(seed), no, npv, na, ah, ch, 1.0)
    ac = ActorCritic(no, npv, na, actor_hidden_dims=list(ah), critic_hidden_dims=list(ch), init_noise_std=1.0)
    ac.load_state_dict(mk().state_dict())
    alg = PPO(ac, **kw)
    alg.init_storage(N, T, [no], [npv], [na])
    orc = PPOOracle(mk(), N, T, **kw)
    inp = rollout_inputs(seed, T, N, no, npv, na)
    for t in range(T):
        alg.act(inp["obs"][t], inp["priv"][t], eps=inp["eps"][t])
        orc.act(inp["obs"][t], inp["priv"][t], inp["eps"][t])
        alg.process_env_step(inp["rewards"][t], inp["dones"][t].astype(np.uint8), {"time_outs": inp["time_outs"][t].astype(np.uint8)})
        orc.process_env_step(inp["rewards"][t], inp["dones"][t], inp["time_outs"][t])
    alg.compute_returns(inp["priv"][T])
    orc.compute_returns(inp["priv"][T])
    np.testing.assert_allclose(alg.buffer(5, (T, N)).numpy(), orc.returns, rtol=1e-4, atol=1e-4)
    perm = np.random.default_rng(4).permutation(T * N).astype(np.int32)
    mvl, msl = alg.update(perm=perm)
    ovl, osl = orc.update(perm)
    assert abs(mvl - ovl) <= 1e-4 * max(1.0, abs(ovl)) and abs(msl - osl) <= 1e-4
    assert alg.learning_rate == pytest.approx(lr) and orc.lr == lr            # fixed schedule
    assert min(orc.gnorm_hist) > 0.05                                          # every step was clipped
    sd = ac.state_dict()
    for k, v in orc.ac.state_dict().items():
        d = np.abs(sd[k] - v)
        assert d.max() <= 2 * epochs * nmb * lr, (k, d.max())
        assert np.mean(d > 3e-6) < 2e-3, (k, float(np.mean(d > 3e-6)))
