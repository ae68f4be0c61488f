"""CPU: the compiled host restatement of the learner's dense arithmetic (oracle/host/hx_learner_host.cpp, what bench.py's
cpu_baseline times) against numpy and against the golden outputs of the reference's own PPO (tests/golden/ppo_small.npz)."""
import os

import numpy as np
import pytest

from oracle.host.learner import HostMLP, HostPPOOracle, host_actor_critic
from oracle.ppo import MLP, ActorCriticOracle
from tests.ppo_inputs import rollout_inputs

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("M,dims", [(37, [23, 17, 9, 5]), (500, [615, 64, 32, 10]), (193, [70, 128, 6, 1])])
def test_compiled_mlp_equals_numpy(M, dims):
    rng = np.random.default_rng(M)
    Ws = [rng.standard_normal((dims[i + 1], dims[i])).astype(np.float32) / np.sqrt(dims[i]) for i in range(len(dims) - 1)]
    bs = [rng.standard_normal(dims[i + 1]).astype(np.float32) for i in range(len(dims) - 1)]
    x, dout = rng.standard_normal((M, dims[0])).astype(np.float32), rng.standard_normal((M, dims[-1])).astype(np.float32)
    ref, host = MLP(Ws, bs), HostMLP(Ws, bs)
    y0, hs0 = ref.forward(x, keep=True)
    y1, hs1 = host.forward(x, keep=True)
    np.testing.assert_allclose(y1, y0, rtol=1e-5, atol=1e-5)
    dW0, db0 = ref.backward(hs0, dout)
    dW1, db1 = host.backward(hs1, dout)
    for a, b in zip(dW0 + db0, dW1 + db1):
        np.testing.assert_allclose(b, a, rtol=2e-4, atol=2e-4 * max(1.0, float(np.abs(a).max())))


def test_compiled_learner_reproduces_reference_fixture():
    """The whole PPO iteration on the compiled learner against the reference's own outputs (same checks as
    tests/test_oracle_ppo.py; tolerances those of an fp32 GEMM with another summation order)."""
    fx = np.load(os.path.join(GOLD, "ppo_small.npz"))
    seed, T, N, ep, nmb = (int(x) for x in fx["meta"])
    ac = host_actor_critic(ActorCriticOracle.default_init(np.random.default_rng(seed)))
    init = {k: v.copy() for k, v in ac.state_dict().items()}
    alg = HostPPOOracle(ac, N, T, num_learning_epochs=ep, num_mini_batches=nmb, learning_rate=float(fx["lr0"]))
    inp = rollout_inputs(seed, T, N)
    for t in range(T):
        a = alg.act(inp["obs"][t], inp["priv"][t], inp["eps"][t])
        np.testing.assert_allclose(a, fx["actions"][t], rtol=0, atol=5e-6)
        np.testing.assert_allclose(alg._tr["v"], fx["values"][t], rtol=0, atol=5e-6)
        alg.process_env_step(inp["rewards"][t] * np.float32(fx["scale_rewards"]), inp["dones"][t], inp["time_outs"][t])
    alg.compute_returns(inp["priv"][T])
    np.testing.assert_allclose(alg.returns, fx["returns"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(alg.advantages, fx["advantages"], rtol=2e-5, atol=5e-5)
    mvl, msl = alg.update(fx["perm"])
    assert abs(mvl - float(fx["mean_value_loss"])) < 2e-5 * max(1, abs(mvl))
    assert abs(msl - float(fx["mean_surrogate_loss"])) < 2e-5
    np.testing.assert_allclose(alg.lr_hist, fx["lrs"], rtol=1e-12)
    np.testing.assert_allclose(alg.gnorm_hist, fx["grad_norms"], rtol=2e-4)
    sd = ac.state_dict()
    for k in sd:
        d = sd[k].astype(np.float64) - init[k]
        assert abs(np.abs(d).sum() - float(fx["delta_abs_" + k])) <= 5e-4 * float(fx["delta_abs_" + k]) + 1e-12, k
        np.testing.assert_allclose(sd[k].reshape(-1)[:64], fx["slice_" + k], rtol=0, atol=1e-6)
