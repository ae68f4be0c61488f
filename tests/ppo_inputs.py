"""Deterministic inputs shared by tests/golden/make_ppo_fixtures.py (which feeds them to the reference's
PPO) and the parity tests (which feed them to the oracle and to the HIP learner).  Regenerated from a
seed instead of being stored, so the committed fixtures hold only the reference's OUTPUTS."""
import numpy as np

F = np.float32


def rollout_inputs(seed, T, N, num_obs=615, num_priv=1050, num_actions=10):
    r = np.random.default_rng(seed)
    d = dict(
        obs=r.standard_normal((T + 1, N, num_obs)).astype(F),
        priv=r.standard_normal((T + 1, N, num_priv)).astype(F),
        eps=r.standard_normal((T, N, num_actions)).astype(F),
        rewards=r.uniform(0, 0.05, (T, N)).astype(F),
        dones=(r.uniform(size=(T, N)) < 0.08),
        time_outs=(r.uniform(size=(T, N)) < 0.05),
    )
    d["time_outs"] &= d["dones"]
    return d
