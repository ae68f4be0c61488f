#!/usr/bin/env python3
"""Generate tests/golden/ppo_*.npz by running the REFERENCE learner.

What runs: the reference's unmodified humanoid/algo/ppo/{actor_critic,rollout_storage,ppo}.py imported
from /root/reference (tests/refstub/loader.py::load_ppo), torch CPU fp32, hector dimensions
(615 / 1050 / 10, actor [512,256,128], critic [768,256,128]; reference hector_config.py:207-219).
Weights and inputs are regenerated from seeds (oracle.ppo.ActorCriticOracle.default_init,
tests/ppo_inputs.py); the two random draws inside the reference (Normal.sample, torch.randperm) are
replaced by / recorded as injected values.  Stored: the reference's outputs only.

Run in this container only:  python tests/golden/make_ppo_fixtures.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
from tests.refstub import loader  # noqa: E402
from tests.ppo_inputs import rollout_inputs  # noqa: E402
from oracle.ppo import ActorCriticOracle  # noqa: E402


def generate(name, seed, T, N, lr, epochs=2, nmb=4, scale_rewards=1.0):
    ac_mod, rs_mod, ppo_mod = loader.load_ppo()
    torch.manual_seed(seed)
    torch.set_num_threads(8)
    init = ActorCriticOracle.default_init(np.random.default_rng(seed))
    ac = ac_mod.ActorCritic(615, 1050, 10, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[768, 256, 128],
                            init_noise_std=1.0)
    ac.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in init.state_dict().items()})
    alg = ppo_mod.PPO(ac, num_learning_epochs=epochs, num_mini_batches=nmb, clip_param=0.2, gamma=0.994, lam=0.9,
                      value_loss_coef=1.0, entropy_coef=0.001, learning_rate=lr, max_grad_norm=1.0,
                      use_clipped_value_loss=True, schedule="adaptive", desired_kl=0.01, device="cpu")
    alg.init_storage(N, T, [615], [1050], [10])
    inp = rollout_inputs(seed, T, N)

    eps_box = {}
    real_normal, real_randperm = torch.normal, torch.randperm
    torch.normal = lambda mean, std, **k: mean + std * eps_box["eps"]
    perm_box = {}

    def randperm(n, **k):
        p = real_randperm(n, **k)
        perm_box["perm"] = p.clone()
        return p
    torch.randperm = randperm
    out = {}
    try:
        acts, vals, logps, mus = [], [], [], []
        with torch.inference_mode():
            for t in range(T):
                eps_box["eps"] = torch.from_numpy(inp["eps"][t])
                a = alg.act(torch.from_numpy(inp["obs"][t]), torch.from_numpy(inp["priv"][t]))
                acts.append(a.numpy().copy())
                vals.append(alg.transition.values.numpy()[:, 0].copy())
                logps.append(alg.transition.actions_log_prob.numpy().copy())
                mus.append(alg.transition.action_mean.numpy().copy())
                alg.process_env_step(torch.from_numpy(inp["rewards"][t] * np.float32(scale_rewards)),
                                     torch.from_numpy(inp["dones"][t]), {"time_outs": torch.from_numpy(inp["time_outs"][t])})
            alg.compute_returns(torch.from_numpy(inp["priv"][T]))
        out.update(actions=np.stack(acts), values=np.stack(vals), logp=np.stack(logps), mu=np.stack(mus),
                   stored_rewards=alg.storage.rewards.numpy()[..., 0].copy(),
                   returns=alg.storage.returns.numpy()[..., 0].copy(),
                   advantages=alg.storage.advantages.numpy()[..., 0].copy())
        # record per-minibatch internals by wrapping the optimizer step
        gn, kls, lrs = [], [], []
        real_clip = torch.nn.utils.clip_grad_norm_

        def clip(params, max_norm):
            params = list(params)
            n = real_clip(params, max_norm)
            gn.append(float(n))
            lrs.append(alg.learning_rate)
            return n
        ppo_mod.nn.utils.clip_grad_norm_ = clip
        torch.normal = real_normal      # update() samples once per minibatch and discards the draw
        mvl, msl = alg.update()
        ppo_mod.nn.utils.clip_grad_norm_ = real_clip
        out.update(mean_value_loss=mvl, mean_surrogate_loss=msl, grad_norms=np.array(gn), lrs=np.array(lrs),
                   perm=perm_box["perm"].numpy().astype(np.int32), final_lr=alg.learning_rate)
        sd = {k: v.detach().numpy() for k, v in ac.state_dict().items()}
        init_sd = init.state_dict()
        for k, v in sd.items():
            d = (v.astype(np.float64) - init_sd[k].astype(np.float64))
            out["delta_sum_" + k] = d.sum()
            out["delta_abs_" + k] = np.abs(d).sum()
            out["slice_" + k] = v.reshape(-1)[:64].copy()
            out["slice_end_" + k] = v.reshape(-1)[-64:].copy()
        st = alg.optimizer.state_dict()["state"]
        out["adam_m_std"] = st[0]["exp_avg"].numpy().copy()
        out["adam_v_std"] = st[0]["exp_avg_sq"].numpy().copy()
        out["adam_m_actor0_slice"] = st[1]["exp_avg"].numpy().reshape(-1)[:64].copy()
        out["adam_v_actor0_slice"] = st[1]["exp_avg_sq"].numpy().reshape(-1)[:64].copy()
        out["adam_step"] = float(st[0]["step"])
        out["meta"] = np.array([seed, T, N, epochs, nmb], np.int64)
        out["lr0"] = lr
        out["scale_rewards"] = scale_rewards
    finally:
        torch.normal, torch.randperm = real_normal, real_randperm
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(name, "vloss %.5f sloss %.5f lrs" % (mvl, msl), np.round(np.array(lrs) * 1e5, 3), "gn", np.round(gn, 3),
          "size %.0f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    assert loader.available()
    generate("ppo_small", seed=11, T=6, N=16, lr=1e-5)
    # larger lr and reward scale: makes the KL schedule move in both directions and clipping active
    generate("ppo_clip", seed=12, T=8, N=32, lr=1e-3, scale_rewards=40.0)
