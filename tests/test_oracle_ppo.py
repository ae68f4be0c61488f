"""CPU: the learner oracle (oracle/ppo.py) against the golden outputs of the reference's own PPO
(tests/golden/make_ppo_fixtures.py).  Pins SURVEY.md 8a rows a12-a15."""
import os

import numpy as np
import pytest

from oracle.ppo import ActorCriticOracle, PPOOracle
from tests.ppo_inputs import rollout_inputs

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name", ["ppo_small", "ppo_clip"])
def test_oracle_ppo_reproduces_reference(name):
    fx = np.load(os.path.join(GOLD, name + ".npz"))
    seed, T, N, ep, nmb = (int(x) for x in fx["meta"])
    ac = ActorCriticOracle.default_init(np.random.default_rng(seed))
    init = {k: v.copy() for k, v in ac.state_dict().items()}
    alg = PPOOracle(ac, N, T, num_learning_epochs=ep, num_mini_batches=nmb, learning_rate=float(fx["lr0"]))
    inp = rollout_inputs(seed, T, N)
    for t in range(T):
        a = alg.act(inp["obs"][t], inp["priv"][t], inp["eps"][t])
        np.testing.assert_allclose(a, fx["actions"][t], rtol=0, atol=2e-6)
        np.testing.assert_allclose(alg._tr["v"], fx["values"][t], rtol=0, atol=2e-6)
        np.testing.assert_allclose(alg._tr["logp"], fx["logp"][t], rtol=0, atol=2e-5)
        alg.process_env_step(inp["rewards"][t] * np.float32(fx["scale_rewards"]), inp["dones"][t], inp["time_outs"][t])
    alg.compute_returns(inp["priv"][T])
    np.testing.assert_allclose(alg.rewards, fx["stored_rewards"], rtol=0, atol=1e-6)     # time-out bootstrap
    np.testing.assert_allclose(alg.returns, fx["returns"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(alg.advantages, fx["advantages"], rtol=1e-5, atol=1e-5)
    mvl, msl = alg.update(fx["perm"])
    assert abs(mvl - float(fx["mean_value_loss"])) < 1e-5 * max(1, abs(mvl))
    assert abs(msl - float(fx["mean_surrogate_loss"])) < 1e-5
    np.testing.assert_allclose(alg.lr_hist, fx["lrs"], rtol=1e-12)               # adaptive-KL schedule, both directions
    np.testing.assert_allclose(alg.gnorm_hist, fx["grad_norms"], rtol=1e-4)
    sd = ac.state_dict()
    for k in sd:
        d = sd[k].astype(np.float64) - init[k]
        assert abs(np.abs(d).sum() - float(fx["delta_abs_" + k])) <= 2e-4 * float(fx["delta_abs_" + k]) + 1e-12, k
        np.testing.assert_allclose(sd[k].reshape(-1)[:64], fx["slice_" + k], rtol=0, atol=5e-7)
        np.testing.assert_allclose(sd[k].reshape(-1)[-64:], fx["slice_end_" + k], rtol=0, atol=5e-7)
    np.testing.assert_allclose(alg.m[0], fx["adam_m_std"], rtol=1e-3, atol=1e-7)
    np.testing.assert_allclose(alg.v[0], fx["adam_v_std"], rtol=2e-3, atol=1e-10)
    assert alg.t == int(fx["adam_step"])


def test_schedule_moves_both_ways():
    up, down = np.load(os.path.join(GOLD, "ppo_small.npz")), np.load(os.path.join(GOLD, "ppo_clip.npz"))
    assert up["lrs"][-1] > up["lrs"][0] and down["lrs"][-1] < down["lrs"][0]


def test_rollout_overflow():
    ac = ActorCriticOracle.default_init(np.random.default_rng(0), 8, 8, 2, (4, 4, 4), (4, 4, 4))
    alg = PPOOracle(ac, 2, 1)
    alg.act(np.zeros((2, 8)), np.zeros((2, 8)), np.zeros((2, 2)))
    alg.process_env_step(np.zeros(2), np.zeros(2, bool))
    with pytest.raises(AssertionError, match="Rollout buffer overflow"):
        alg.process_env_step(np.zeros(2), np.zeros(2, bool))
