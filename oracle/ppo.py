"""ORACLE (test infrastructure, never shipped) -- CPU restatement of the reference PPO learner.

Follows the reference's
  humanoid/algo/ppo/actor_critic.py    (ActorCritic :36-128: two ELU MLPs, state-independent std, Normal)
  humanoid/algo/ppo/rollout_storage.py (add_transitions :87-100, compute_returns :122-136,
                                        mini_batch_generator :146-182)
  humanoid/algo/ppo/ppo.py             (act :91-101, process_env_step :103-113, compute_returns :115-117,
                                        update :119-184)
with the autograd graph written out by hand (the backward formulas are what the HIP kernels implement).
float32 numpy; GEMMs go through numpy's BLAS.

Pinning: tests/test_oracle_ppo.py compares every stage against tests/golden/ppo_*.npz, produced by
importing the reference's own modules (tests/golden/make_ppo_fixtures.py).

Random numbers are injected: `eps` for action sampling (reference: Normal.sample, actor_critic.py:117)
and `perm` for the minibatch permutation (reference: torch.randperm, rollout_storage.py:149).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np

F = np.float32
LOG_SQRT_2PI = F(0.5 * np.log(2 * np.pi))


def elu(z):
    return np.where(z > 0, z, np.expm1(np.minimum(z, 0))).astype(F)


def bf16_round(x):
    """fp32 -> bf16 (round to nearest even) -> fp32: the rounding the product's bf16 mode applies to the operands of its
    hidden-layer forward and dgrad products (no reference counterpart -- the reference is fp32; BASELINE config 4)."""
    u = np.ascontiguousarray(x, F).view(np.uint32)
    r = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)
    return r.view(F)


class MLP:
    """weights[i]: [out,in] (torch nn.Linear layout), biases[i]: [out].
    bf16=True emulates the product's mixed-precision mode: operands of the HIDDEN layers' forward products and of the
    dgrads / wgrads of hidden layers are rounded to bf16; accumulation, bias (and its gradient), ELU and the output layer stay fp32."""

    def __init__(self, weights, biases):
        self.W = [np.asarray(w, F).copy() for w in weights]
        self.b = [np.asarray(b, F).copy() for b in biases]

    def forward(self, x, keep=False, bf16=False):
        hs = [np.asarray(x, F)]
        h = hs[0]
        L = len(self.W)
        for i in range(L):
            if bf16 and i < L - 1:
                z = (bf16_round(h) @ bf16_round(self.W[i]).T + self.b[i]).astype(F)
            else:
                z = (h @ self.W[i].T + self.b[i]).astype(F)
            h = elu(z) if i < L - 1 else z
            hs.append(h)
        return (h, hs) if keep else h

    def backward(self, hs, dout, bf16=False):
        """dout = dLoss/d(output).  Returns (dW list, db list)."""
        L = len(self.W)
        dW, db = [None] * L, [None] * L
        dz = np.asarray(dout, F)
        for i in range(L - 1, -1, -1):
            if bf16 and i < L - 1:
                dW[i] = (bf16_round(dz).T @ bf16_round(hs[i])).astype(F)     # hidden-layer wgrads: rounded operands
            else:
                dW[i] = (dz.T @ hs[i]).astype(F)
            db[i] = dz.sum(0).astype(F)                                      # bias gradients stay exact fp32 sums
            if i > 0:
                if bf16 and i < L - 1:
                    dh = (bf16_round(dz) @ bf16_round(self.W[i])).astype(F)
                else:
                    dh = (dz @ self.W[i]).astype(F)
                h = hs[i]
                dz = (dh * np.where(h > 0, F(1), h + F(1))).astype(F)
        return dW, db


class ActorCriticOracle:
    def __init__(self, actor_w, actor_b, critic_w, critic_b, std):
        self.actor = MLP(actor_w, actor_b)
        self.critic = MLP(critic_w, critic_b)
        self.std = np.asarray(std, F).copy()

    @staticmethod
    def default_init(rng, num_obs=615, num_priv=1050, num_actions=10, actor_hidden=(512, 256, 128),
                     critic_hidden=(768, 256, 128), init_noise_std=1.0):
        """nn.Linear default init (kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight
        and bias), drawn from a numpy Generator so fixtures need not store 6 MB of weights."""
        def lin(i, o):
            k = 1.0 / np.sqrt(i)
            return rng.uniform(-k, k, (o, i)).astype(F), rng.uniform(-k, k, o).astype(F)
        dims_a = [num_obs, *actor_hidden, num_actions]
        dims_c = [num_priv, *critic_hidden, 1]
        aw, ab = zip(*[lin(dims_a[i], dims_a[i + 1]) for i in range(len(dims_a) - 1)])
        cw, cb = zip(*[lin(dims_c[i], dims_c[i + 1]) for i in range(len(dims_c) - 1)])
        return ActorCriticOracle(aw, ab, cw, cb, init_noise_std * np.ones(num_actions, F))

    def params(self):
        """flat list in torch `parameters()` order: std, actor.{0,2,4,6}.{weight,bias}, critic.*"""
        out = [self.std]
        for net in (self.actor, self.critic):
            for w, b in zip(net.W, net.b):
                out += [w, b]
        return out

    def state_dict(self):
        d = {"std": self.std}
        for name, net in (("actor", self.actor), ("critic", self.critic)):
            for i, (w, b) in enumerate(zip(net.W, net.b)):
                d[f"{name}.{2 * i}.weight"] = w
                d[f"{name}.{2 * i}.bias"] = b
        return d

    def act(self, obs, eps, bf16=False):
        mu = self.actor.forward(obs, bf16=bf16)
        sigma = (mu * F(0) + self.std).astype(F)
        a = (mu + sigma * np.asarray(eps, F)).astype(F)
        return a, mu, sigma

    def log_prob(self, a, mu, sigma):
        var = sigma * sigma
        return np.sum(-((a - mu) ** 2) / (F(2) * var) - np.log(sigma) - LOG_SQRT_2PI, -1).astype(F)

    def evaluate(self, priv, bf16=False):
        return self.critic.forward(priv, bf16=bf16)


class PPOOracle:
    def __init__(self, ac, num_envs, num_steps, num_learning_epochs=2, num_mini_batches=4, clip_param=0.2,
                 gamma=0.994, lam=0.9, value_loss_coef=1.0, entropy_coef=0.001, learning_rate=1e-5,
                 max_grad_norm=1.0, use_clipped_value_loss=True, schedule="adaptive", desired_kl=0.01, bf16=False):
        """bf16=True: emulate the product's mixed-precision mode (hx_ppo_set_compute_dtype 1): every hidden-layer forward
        (rollout actor and critic included), dgrad and wgrad product uses bf16-rounded operands."""
        self.bf16 = bf16
        self.ac = ac
        self.N, self.T = num_envs, num_steps
        self.epochs, self.nmb = num_learning_epochs, num_mini_batches
        self.clip, self.gamma, self.lam = F(clip_param), F(gamma), F(lam)
        self.vcoef, self.ecoef = F(value_loss_coef), F(entropy_coef)
        self.lr = learning_rate
        self.max_grad_norm = max_grad_norm
        self.use_clipped_value_loss = use_clipped_value_loss
        self.schedule, self.desired_kl = schedule, desired_kl
        no = ac.actor.W[0].shape[1]
        npv = ac.critic.W[0].shape[1]
        na = ac.std.shape[0]
        T, N = self.T, self.N
        self.obs = np.zeros((T, N, no), F)
        self.priv = np.zeros((T, N, npv), F)
        self.actions = np.zeros((T, N, na), F)
        self.rewards = np.zeros((T, N), F)
        self.dones = np.zeros((T, N), np.uint8)
        self.values = np.zeros((T, N), F)
        self.logp = np.zeros((T, N), F)
        self.mu = np.zeros((T, N, na), F)
        self.sigma = np.zeros((T, N, na), F)
        self.returns = np.zeros((T, N), F)
        self.advantages = np.zeros((T, N), F)
        self.step = 0
        p = ac.params()
        self.m = [np.zeros_like(x) for x in p]
        self.v = [np.zeros_like(x) for x in p]
        self.t = 0
        self.kl_hist, self.lr_hist = [], []

    # ---- rollout side (ppo.py:91-113)
    def act(self, obs, priv, eps):
        a, mu, sigma = self.ac.act(obs, eps, bf16=self.bf16)
        v = self.ac.evaluate(priv, bf16=self.bf16)[:, 0]
        self._tr = dict(obs=np.asarray(obs, F), priv=np.asarray(priv, F), a=a, v=v,
                        logp=self.ac.log_prob(a, mu, sigma), mu=mu, sigma=sigma)
        return a

    def process_env_step(self, rewards, dones, time_outs=None):
        if self.step >= self.T:
            raise AssertionError("Rollout buffer overflow")
        r = np.asarray(rewards, F).copy()
        if time_outs is not None:
            r = (r + self.gamma * (self._tr["v"] * np.asarray(time_outs).astype(F))).astype(F)
        s, tr = self.step, self._tr
        self.obs[s], self.priv[s], self.actions[s] = tr["obs"], tr["priv"], tr["a"]
        self.rewards[s], self.dones[s] = r, np.asarray(dones).astype(np.uint8)
        self.values[s], self.logp[s], self.mu[s], self.sigma[s] = tr["v"], tr["logp"], tr["mu"], tr["sigma"]
        self.step += 1

    # ---- GAE (rollout_storage.py:122-136)
    def compute_returns(self, last_priv):
        last_values = self.ac.evaluate(last_priv, bf16=self.bf16)[:, 0]
        adv = np.zeros(self.N, F)
        for s in reversed(range(self.T)):
            nv = last_values if s == self.T - 1 else self.values[s + 1]
            nt = F(1.0) - self.dones[s].astype(F)
            delta = self.rewards[s] + nt * self.gamma * nv - self.values[s]
            adv = (delta + nt * self.gamma * self.lam * adv).astype(F)
            self.returns[s] = adv + self.values[s]
        a = (self.returns - self.values).astype(F)
        mean = a.mean(dtype=np.float64)
        std = a.std(ddof=1, dtype=np.float64)          # torch.std default: unbiased
        self.advantages = ((a - F(mean)) / (F(std) + F(1e-8))).astype(F)

    # ---- update (ppo.py:119-184)
    def loss_and_grads(self, idx):
        """One minibatch: returns dict(loss terms, kl_mean) and gradients in params() order."""
        ac = self.ac
        M = len(idx)
        flat = lambda x: x.reshape((self.T * self.N,) + x.shape[2:])
        obs, priv = flat(self.obs)[idx], flat(self.priv)[idx]
        a, v_old, ret = flat(self.actions)[idx], flat(self.values)[idx], flat(self.returns)[idx]
        logp_old, adv = flat(self.logp)[idx], flat(self.advantages)[idx]
        mu_old, sig_old = flat(self.mu)[idx], flat(self.sigma)[idx]

        mu, hs_a = ac.actor.forward(obs, keep=True, bf16=self.bf16)
        sigma = (mu * F(0) + ac.std).astype(F)
        logp = ac.log_prob(a, mu, sigma)
        v, hs_c = ac.critic.forward(priv, keep=True, bf16=self.bf16)
        v = v[:, 0]
        entropy = np.sum(F(0.5) + LOG_SQRT_2PI + np.log(sigma), -1).astype(F)

        kl = np.sum(np.log(sigma / sig_old + F(1e-5)) + (sig_old ** 2 + (mu_old - mu) ** 2) / (F(2) * sigma ** 2) - F(0.5), -1)
        kl_mean = float(np.mean(kl.astype(F)))

        ratio = np.exp(logp - logp_old).astype(F)
        s = -adv * ratio
        lo, hi = F(1) - self.clip, F(1) + self.clip
        sc = -adv * np.clip(ratio, lo, hi)
        surrogate_loss = float(np.mean(np.maximum(s, sc)))
        inr = ((ratio >= lo) & (ratio <= hi)).astype(F)
        w = np.where(s > sc, F(1), np.where(s == sc, F(0.5) + F(0.5) * inr, inr))
        dlogp = (-adv * w * ratio / F(M)).astype(F)

        if self.use_clipped_value_loss:
            vc = v_old + np.clip(v - v_old, -self.clip, self.clip)
            la, lb = (v - ret) ** 2, (vc - ret) ** 2
            value_loss = float(np.mean(np.maximum(la, lb)))
            inv = (np.abs(v - v_old) <= self.clip).astype(F)
            ga, gb = F(2) * (v - ret), F(2) * (vc - ret) * inv
            dv = np.where(la > lb, ga, np.where(la == lb, F(0.5) * ga + F(0.5) * gb, gb))
        else:
            value_loss = float(np.mean((ret - v) ** 2))
            dv = F(2) * (v - ret)
        dv = (self.vcoef * dv / F(M)).astype(F)

        dmu = (dlogp[:, None] * (a - mu) / (sigma * sigma)).astype(F)
        dsig = dlogp[:, None] * ((a - mu) ** 2 / sigma ** 3 - F(1) / sigma) - self.ecoef / (F(M) * sigma)
        dstd = dsig.sum(0).astype(F)
        dWa, dba = ac.actor.backward(hs_a, dmu, bf16=self.bf16)
        dWc, dbc = ac.critic.backward(hs_c, dv[:, None], bf16=self.bf16)
        grads = [dstd]
        for dW, db in ((dWa, dba), (dWc, dbc)):
            for x, y in zip(dW, db):
                grads += [x, y]
        loss = surrogate_loss + float(self.vcoef) * value_loss - float(self.ecoef) * float(entropy.mean())
        return dict(loss=loss, surrogate=surrogate_loss, value=value_loss, kl=kl_mean,
                    entropy=float(entropy.mean())), grads

    def adapt_lr(self, kl_mean):
        if self.desired_kl is not None and self.schedule == "adaptive":
            if kl_mean > self.desired_kl * 2.0:
                self.lr = max(1e-5, self.lr / 1.5)
            elif kl_mean < self.desired_kl / 2.0 and kl_mean > 0.0:
                self.lr = min(1e-2, self.lr * 1.5)

    def optimizer_step(self, grads):
        total = np.sqrt(sum(float(np.sum(g.astype(np.float64) ** 2)) for g in grads))
        coef = min(1.0, self.max_grad_norm / (total + 1e-6))
        self.t += 1
        b1, b2, eps = 0.9, 0.999, 1e-8
        bc1, bc2 = 1 - b1 ** self.t, 1 - b2 ** self.t
        for p, g, m, v in zip(self.ac.params(), grads, self.m, self.v):
            g = (g * F(coef)).astype(F)
            m[...] = F(b1) * m + F(1 - b1) * g
            v[...] = F(b2) * v + F(1 - b2) * g * g
            denom = np.sqrt(v) / F(np.sqrt(bc2)) + F(eps)
            p[...] = p - F(self.lr / bc1) * (m / denom)
        return total

    def update(self, perm):
        """perm: permutation of T*N indices (reference draws it once and reuses it for every epoch)."""
        B = self.T * self.N
        mbs = B // self.nmb
        sv = ss = 0.0
        self.kl_hist, self.lr_hist, self.gnorm_hist = [], [], []
        for _ in range(self.epochs):
            for i in range(self.nmb):
                idx = np.asarray(perm[i * mbs:(i + 1) * mbs])
                info, grads = self.loss_and_grads(idx)
                self.adapt_lr(info["kl"])
                self.gnorm_hist.append(self.optimizer_step(grads))
                self.kl_hist.append(info["kl"])
                self.lr_hist.append(self.lr)
                sv += info["value"]
                ss += info["surrogate"]
        n = self.epochs * self.nmb
        self.step = 0
        return sv / n, ss / n
