"""ActorCritic / PPO host objects with the reference's (rsl_rl-style) surface, served by libhx.so.

  ActorCritic   humanoid/algo/ppo/actor_critic.py:36-128   (state_dict keys: std, actor.{0,2,4,6}.*, critic.*)
  PPO           humanoid/algo/ppo/ppo.py:38-184            (init_storage / act / process_env_step /
                                                            compute_returns / update, .learning_rate, .storage)
The two reference objects share one device object here (parameters, rollout storage and workspace live in
one hx_ppo handle), because the learner's kernels need them resident together; `ActorCritic` only records
the architecture until `PPO.init_storage` binds it.  No arithmetic happens in this file.
"""
from collections import OrderedDict

import numpy as np

from .. import capi
from ..devarray import DeviceArray, device_pointer


class ActorCritic:
    is_recurrent = False

    def __init__(self, num_actor_obs, num_critic_obs, num_actions, actor_hidden_dims=(256, 256, 256),
                 critic_hidden_dims=(256, 256, 256), init_noise_std=1.0, activation="elu", **kwargs):
        if kwargs:
            print("ActorCritic.__init__ got unexpected arguments, which will be ignored: " + str(list(kwargs)))
        if len(actor_hidden_dims) != 3 or len(critic_hidden_dims) != 3:
            raise ValueError("the HIP learner is built for three hidden layers per network (hector: [512,256,128] / [768,256,128])")
        self.num_actor_obs, self.num_critic_obs, self.num_actions = num_actor_obs, num_critic_obs, num_actions
        self.actor_hidden_dims, self.critic_hidden_dims = list(actor_hidden_dims), list(critic_hidden_dims)
        self.init_noise_std = init_noise_std
        self._alg = None
        self._pending_state = self._default_init()

    # ---- parameter bookkeeping (torch `parameters()` order)
    def tensor_shapes(self):
        shapes = OrderedDict(std=(self.num_actions,))
        for name, dims in (("actor", [self.num_actor_obs, *self.actor_hidden_dims, self.num_actions]),
                           ("critic", [self.num_critic_obs, *self.critic_hidden_dims, 1])):
            for i in range(4):
                shapes[f"{name}.{2 * i}.weight"] = (dims[i + 1], dims[i])
                shapes[f"{name}.{2 * i}.bias"] = (dims[i + 1],)
        return shapes

    def num_params(self):
        return int(sum(int(np.prod(s)) for s in self.tensor_shapes().values()))

    def _default_init(self):
        """nn.Linear default init: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight and bias
        (reference uses the torch defaults: `init_weights` is unused, actor_critic.py:86-90).
        Drawn from torch's generator when available so `torch.manual_seed(s)` gives the reference's stream order;
        the values differ from torch's own kaiming call sequence only in draw order."""
        out = OrderedDict()
        try:
            import torch
            uni = lambda shape, k: ((torch.rand(*shape) * 2 - 1) * k).numpy().astype(np.float32)
        except ImportError:                              # pragma: no cover
            uni = lambda shape, k: np.random.uniform(-k, k, shape).astype(np.float32)
        for name, shape in self.tensor_shapes().items():
            if name == "std":
                out[name] = np.full(shape, self.init_noise_std, np.float32)
            else:
                fan_in = shape[1] if len(shape) == 2 else self._fan_in_of_bias(name)
                out[name] = uni(shape, 1.0 / np.sqrt(fan_in))
        return out

    def _fan_in_of_bias(self, name):
        return self.tensor_shapes()[name.replace("bias", "weight")][1]

    def _flatten(self, sd):
        parts = []
        for name, shape in self.tensor_shapes().items():
            t = sd[name]
            t = t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)
            if tuple(t.shape) != tuple(shape):
                raise ValueError(f"state_dict[{name!r}] has shape {tuple(t.shape)}, expected {shape}")
            parts.append(np.ascontiguousarray(t, np.float32).reshape(-1))
        return np.concatenate(parts)

    def _unflatten(self, flat):
        out, o = OrderedDict(), 0
        for name, shape in self.tensor_shapes().items():
            n = int(np.prod(shape))
            out[name] = flat[o:o + n].reshape(shape).copy()
            o += n
        return out

    def state_dict(self):
        if self._alg is None:
            return OrderedDict((k, v.copy()) for k, v in self._pending_state.items())
        flat = np.empty(self.num_params(), np.float32)
        capi.check(capi.lib().hx_ppo_get_params_h(self._alg._h, capi.ptr(flat)), "get_params")
        return self._unflatten(flat)

    def load_state_dict(self, sd, strict=True):
        if self._alg is None:
            self._pending_state = self._unflatten(self._flatten(sd))
        else:
            flat = self._flatten(sd)
            capi.check(capi.lib().hx_ppo_set_params_h(self._alg._h, capi.ptr(flat)), "set_params")

    def load_actor_from_onnx(self, path):
        """Replace the actor's weights with those of an exported actor (reference play.py:89-98 and the shipped
        humanoid/locomotion_net*.onnx); critic and std keep their values."""
        from ..utils.onnx_io import actor_state_dict
        sd = self.state_dict()
        new = actor_state_dict(path)
        for k, v in new.items():
            if k not in sd or sd[k].shape != v.shape:
                raise ValueError(f"{path}: tensor {k} {v.shape} does not fit this actor ({sd.get(k, np.zeros(0)).shape})")
            sd[k] = v
        self.load_state_dict(sd)

    @property
    def std(self):
        return self.state_dict()["std"]

    # ---- modes / no-ops kept for API compatibility
    def train(self):
        return self

    def eval(self):
        return self

    def to(self, device):
        return self

    def reset(self, dones=None):
        pass

    def act_inference(self, observations):
        if self._alg is None:
            raise RuntimeError("ActorCritic is not bound to a learner yet (PPO.init_storage)")
        return self._alg.inference(observations)


class PPO:
    def __init__(self, actor_critic, num_learning_epochs=1, num_mini_batches=1, clip_param=0.2, gamma=0.998, lam=0.95,
                 value_loss_coef=1.0, entropy_coef=0.0, learning_rate=1e-3, max_grad_norm=1.0,
                 use_clipped_value_loss=True, schedule="fixed", desired_kl=0.01, device="cuda:0", stream=None,
                 comm=None, mlp_dtype="f32", seed=None):
        """mlp_dtype: "f32" (default, the reference's precision) or "bf16" (BASELINE config 4: forward / dgrad products of
        the update and the deferred critic on the bf16 matrix cores, fp32 master weights; hx_ppo_set_compute_dtype)."""
        if mlp_dtype not in ("f32", "bf16"):
            raise ValueError("mlp_dtype must be 'f32' or 'bf16'")
        self.mlp_dtype = mlp_dtype
        self.seed = seed                  # None: the library's fixed keys (what the reference-fixture tests were generated with)
        self.device = device
        self.actor_critic = actor_critic
        self.desired_kl, self.schedule = desired_kl, schedule
        self._lr0 = learning_rate
        self.clip_param, self.num_learning_epochs, self.num_mini_batches = clip_param, num_learning_epochs, num_mini_batches
        self.value_loss_coef, self.entropy_coef = value_loss_coef, entropy_coef
        self.gamma, self.lam, self.max_grad_norm = gamma, lam, max_grad_norm
        self.use_clipped_value_loss = use_clipped_value_loss
        self.storage = None
        self._h = None
        self._stream = stream
        self.comm = comm                  # isaac_amd.parallel.Comm or None (single process)
        self._L = capi.lib()
        self.creation_knobs = {}          # HX_* scheduling knobs for this learner's creation (init_storage), e.g. from the runner
        self._keep = []

    # ------------------------------------------------------------------ construction
    def init_storage(self, num_envs, num_transitions_per_env, actor_obs_shape, critic_obs_shape, action_shape,
                     obs_ld=None, priv_ld=None, frames=None):
        """frames = (obs_frame, priv_frame, obs_stack, priv_stack) -- what an env reports as `frame_dims` -- selects single-frame
        observation storage (include/hx_ppo.h hx_ppo_cfg.obs_frame): the rollout storage keeps every robot's frames once and
        the learner's first layers read their rows in place; such a learner is driven by rollout() (hx_rollout), act() and
        process_env_step() with ready-made rows are refused.  None: rows are stored as given (the reference's layout)."""
        ac = self.actor_critic
        c = capi.PpoCfg()
        c.num_envs, c.num_steps = num_envs, num_transitions_per_env
        c.num_obs, c.num_priv, c.num_actions = actor_obs_shape[0], critic_obs_shape[0], action_shape[0]
        for i in range(3):
            c.actor_hidden[i], c.critic_hidden[i] = ac.actor_hidden_dims[i], ac.critic_hidden_dims[i]
        c.num_learning_epochs, c.num_mini_batches = self.num_learning_epochs, self.num_mini_batches
        c.clip_param, c.gamma, c.lam = self.clip_param, self.gamma, self.lam
        c.value_loss_coef, c.entropy_coef = self.value_loss_coef, self.entropy_coef
        c.learning_rate, c.max_grad_norm = self._lr0, self.max_grad_norm
        c.use_clipped_value_loss = int(self.use_clipped_value_loss)
        c.adaptive_schedule = int(self.desired_kl is not None and self.schedule == "adaptive")
        c.desired_kl = self.desired_kl if self.desired_kl is not None else 0.0
        c.init_noise_std = ac.init_noise_std
        c.obs_ld = obs_ld if obs_ld is not None else (c.num_obs + 3) // 4 * 4
        c.priv_ld = priv_ld if priv_ld is not None else (c.num_priv + 3) // 4 * 4
        self.frames = None
        if frames is not None and self.mlp_dtype == "f32":      # the bf16 kernels read ready-made rows
            c.obs_frame, c.priv_frame, c.obs_stack, c.priv_stack = (int(x) for x in frames)
            self.frames = tuple(int(x) for x in frames)
        if (num_envs * num_transitions_per_env) % self.num_mini_batches:
            raise ValueError("num_envs * num_steps_per_env must be divisible by num_mini_batches")
        self._cfg = c
        ext = None
        if self._distributed() and not getattr(self.comm, "in_library", False):
            ext = self.comm.alloc_grad_buffer(self._padded_count(c) + 4)
        h = capi.C.c_void_p()
        # experiment knobs a caller wants for THIS learner (they are environment variables read once, at creation): set for the
        # duration of the call unless the process environment already says otherwise
        import os
        held = {k: v for k, v in self.creation_knobs.items() if k not in os.environ}
        os.environ.update(held)
        try:
            capi.check(self._L.hx_ppo_create(capi.C.byref(c), self._stream, ext, capi.C.byref(h)), "hx_ppo_create")
        finally:
            for k in held:
                del os.environ[k]
        self._h = h
        self._stream = self._L.hx_ppo_stream(h)
        self.N, self.T, self.A = num_envs, num_transitions_per_env, action_shape[0]
        self.obs_ld, self.priv_ld = c.obs_ld, c.priv_ld
        ac._alg = self
        if self.mlp_dtype == "bf16":
            capi.check(self._L.hx_ppo_set_compute_dtype(h, 1), "hx_ppo_set_compute_dtype")
        ac.load_state_dict(ac._pending_state)
        if self._distributed() and getattr(self.comm, "in_library", False):
            self.comm.attach(self)        # RCCL inside the library: hx_ppo_set_comm + parameter broadcast from rank 0
        if self.seed is not None:         # exploration noise keyed per rank, permutation keyed by the run's seed
            rank = self.comm.rank if self.comm is not None else 0
            capi.check(self._L.hx_ppo_set_seed(h, int(self.seed) + rank, int(self.seed)), "hx_ppo_set_seed")
        self.storage = self          # `alg.storage.clear()` style calls land here
        self.step = 0

    def _distributed(self):
        # `force_collectives` lets a single rank walk the multi-rank code path (GPU plumbing test)
        return self.comm is not None and (self.comm.world_size > 1 or getattr(self.comm, "force_collectives", False))

    @staticmethod
    def _padded_count(c):
        r4 = lambda x: (x + 3) // 4 * 4
        tot = 0
        for dims in ([c.num_obs, *c.actor_hidden, c.num_actions], [c.num_priv, *c.critic_hidden, 1]):
            for i in range(4):
                tot = r4(tot + dims[i + 1] * r4(dims[i]))
                tot = r4(tot + dims[i + 1])
        return tot + r4(c.num_actions)

    # ------------------------------------------------------------------ rollout side
    def _padded_ptr(self, x, width, ld):
        """Device pointer of an [N, ld]-strided buffer.  Host arrays of logical width are padded and uploaded."""
        if isinstance(x, DeviceArray) or hasattr(x, "__cuda_array_interface__"):
            p, keep = device_pointer(x)
            return p, keep
        arr = x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)
        arr = np.asarray(arr, np.float32)
        if arr.shape[1] != ld:
            pad = np.zeros((arr.shape[0], ld), np.float32)
            pad[:, :arr.shape[1]] = arr
            arr = pad
        return device_pointer(arr)

    def act(self, obs, critic_obs, eps=None):
        po, k1 = self._padded_ptr(obs, self._cfg.num_obs, self.obs_ld)
        pp, k2 = self._padded_ptr(critic_obs, self._cfg.num_priv, self.priv_ld)
        pe, k3 = (None, None) if eps is None else device_pointer(np.ascontiguousarray(eps, np.float32))
        out = capi.C.c_void_p()
        capi.check(self._L.hx_ppo_act(self._h, po, pp, pe, capi.C.byref(out)), "hx_ppo_act")
        self._keep = [k1, k2, k3]
        return DeviceArray(out.value, (self.N, self.A), np.float32, None, self.stream)

    # ---- shard forms (PipelinedHectorEnv): rows [env0, env0+count) of the current rollout slot, on the shard's stream
    def act_range(self, obs, critic_obs, env0, count, stream):
        po, _ = device_pointer(obs)
        pp, _ = device_pointer(critic_obs)
        out = capi.C.c_void_p()
        capi.check(self._L.hx_ppo_act_range(self._h, po, pp, None, env0, count, stream, capi.C.byref(out)), "hx_ppo_act_range")
        return DeviceArray(out.value, (count, self.A), np.float32, None, stream)

    def process_env_step_range(self, rewards, dones, infos, env0, count, stream, advance):
        pr, _ = device_pointer(rewards)
        pd, _ = device_pointer(dones)
        pt = device_pointer(infos["time_outs"])[0] if "time_outs" in infos else None
        capi.check(self._L.hx_ppo_process_step_range(self._h, pr, pd, pt, env0, count, stream, int(advance)), "hx_ppo_process_step_range")

    def rollout(self, envs, steps):
        """`steps` x {act, env.step, process_env_step} in one C call (hx_rollout).  envs: list of HectorFreeEnv shards
        (one element = the plain unsharded env).  Returns nothing: results live in the rollout storage."""
        n = len(envs)
        sims = (capi.C.c_void_p * n)(*[e._h for e in envs])
        env0 = (capi.C.c_int32 * n)(*[e.env_lo for e in envs])
        cnt = (capi.C.c_int32 * n)(*[e.num_envs for e in envs])
        capi.check(self._L.hx_rollout(self._h, sims, env0, cnt, n, int(steps)), "hx_rollout")
        for e in envs:
            e.common_step_counter += steps
            e._refresh_views()

    def _sync(self, stream):
        """Host wait on a stream that may carry this learner's collectives: through the communicator's deadline when RCCL runs
        inside the library (hx_comm_wait aborts after HX_COMM_TIMEOUT_S), a plain stream synchronisation otherwise."""
        h = getattr(self.comm, "_h", None) if getattr(self.comm, "in_library", False) else None
        if h:
            capi.check(self._L.hx_comm_wait(h, stream, 0.0), "hx_comm_wait")
        else:
            capi.check(self._L.hx_sync(stream), "sync")

    def compute_returns_shards(self, shard_priv):
        """shard_priv: list of (critic_obs, env0, count, stream)."""
        for priv, env0, count, stream in shard_priv:
            capi.check(self._L.hx_ppo_last_values_range(self._h, device_pointer(priv)[0], env0, count, stream), "last_values_range")
            self._sync(stream)
        capi.check(self._L.hx_ppo_compute_returns(self._h, None), "hx_ppo_compute_returns")
        if self._distributed() and not getattr(self.comm, "in_library", False):
            m = capi.C.c_void_p()
            capi.check(self._L.hx_ppo_adv_moments(self._h, capi.C.byref(m)), "adv_moments")
            self.comm.all_reduce_moments(m.value, self.stream)
        capi.check(self._L.hx_ppo_adv_normalize(self._h), "hx_ppo_adv_normalize")

    def process_env_step(self, rewards, dones, infos):
        pr, k1 = device_pointer(rewards)
        pd, k2 = device_pointer(dones if not isinstance(dones, np.ndarray) else dones.astype(np.uint8))
        pt, k3 = (None, None)
        if "time_outs" in infos:
            t = infos["time_outs"]
            pt, k3 = device_pointer(t if not isinstance(t, np.ndarray) else t.astype(np.uint8))
        capi.check(self._L.hx_ppo_process_step(self._h, pr, pd, pt), "hx_ppo_process_step")
        self._keep += [k1, k2, k3]
        self.actor_critic.reset(dones)

    def compute_returns(self, last_critic_obs):
        pp, k = self._padded_ptr(last_critic_obs, self._cfg.num_priv, self.priv_ld)
        capi.check(self._L.hx_ppo_compute_returns(self._h, pp), "hx_ppo_compute_returns")
        if self._distributed() and not getattr(self.comm, "in_library", False):
            m = capi.C.c_void_p()
            capi.check(self._L.hx_ppo_adv_moments(self._h, capi.C.byref(m)), "adv_moments")
            self.comm.all_reduce_moments(m.value, self.stream)
        capi.check(self._L.hx_ppo_adv_normalize(self._h), "hx_ppo_adv_normalize")
        self._keep.append(k)

    # ------------------------------------------------------------------ learner side
    def update(self, perm=None):
        pp, k = (None, None) if perm is None else device_pointer(np.ascontiguousarray(perm, np.int32))
        stats = np.zeros(4, np.float32)
        world = 1 if self.comm is None else self.comm.world_size
        if not self._distributed() or getattr(self.comm, "in_library", False):
            # one C call; with a library-owned communicator it enqueues the RCCL all-reduce of every optimiser step itself
            capi.check(self._L.hx_ppo_update(self._h, pp, capi.ptr(stats)), "hx_ppo_update")
        else:
            capi.check(self._L.hx_ppo_update_begin(self._h, pp), "update_begin")
            g, cnt = capi.C.c_void_p(), capi.C.c_int64()
            for i in range(self.num_learning_epochs * self.num_mini_batches):
                capi.check(self._L.hx_ppo_minibatch_backward(self._h, i, capi.C.byref(g), capi.C.byref(cnt)), "mb_backward")
                self.comm.all_reduce_grads(g.value, cnt.value, self.stream)      # ONE collective per optimiser step
                capi.check(self._L.hx_ppo_minibatch_step(self._h, 1.0 / world), "mb_step")
            capi.check(self._L.hx_ppo_update_end(self._h, capi.ptr(stats)), "update_end")
        self._last_lr, self.last_kl = float(stats[2]), float(stats[3])
        self._keep = []
        return float(stats[0]), float(stats[1])

    @property
    def learning_rate(self):
        lr = capi.C.c_float()
        capi.check(self._L.hx_ppo_get_lr(self._h, capi.C.byref(lr)), "get_lr")
        return float(lr.value)

    @learning_rate.setter
    def learning_rate(self, v):
        capi.check(self._L.hx_ppo_set_lr(self._h, float(v)), "set_lr")

    @property
    def stream(self):
        return self._stream

    def clear(self):
        pass

    def inference(self, obs):
        po, k = self._padded_ptr(obs, self._cfg.num_obs, self.obs_ld)
        rows = obs.shape[0]
        out = capi.DeviceBuffer(rows * self.A * 4)
        capi.check(self._L.hx_ppo_inference(self._h, po, rows, out.ptr), "inference")
        return DeviceArray(out.ptr, (rows, self.A), np.float32, None, self.stream, owner=out)

    def storage_rows(self, which, t0=0, t1=None):
        """RolloutStorage.observations / .privileged_observations of slots [t0, t1) as a [t1 - t0, N, ld] device array, whatever
        the storage layout (hx_ppo_storage_rows; with frame storage t1 may be T + 1)."""
        t1 = self.T if t1 is None else t1
        ld = self.obs_ld if which == capi.PPO_BUF_OBS else self.priv_ld
        out = capi.DeviceBuffer((t1 - t0) * self.N * ld * 4)
        capi.check(self._L.hx_ppo_storage_rows(self._h, which, t0, t1, out.ptr), "hx_ppo_storage_rows")
        return DeviceArray(out.ptr, (t1 - t0, self.N, ld), np.float32, None, self.stream, owner=out)

    def buffer(self, which, shape, dtype=np.float32):
        p = capi.C.c_void_p()
        capi.check(self._L.hx_ppo_buffer(self._h, which, capi.C.byref(p)), "hx_ppo_buffer")
        return DeviceArray(p.value, shape, dtype, None, self.stream)

    def test_mode(self):
        pass

    def train_mode(self):
        pass

    # ---- optimizer state (checkpoint format of on_policy_runner.py:278-295)
    def optimizer_state(self):
        n = self.actor_critic.num_params()
        m, v, step = np.empty(n, np.float32), np.empty(n, np.float32), capi.C.c_int64()
        capi.check(self._L.hx_ppo_get_opt_state_h(self._h, capi.ptr(m), capi.ptr(v), capi.C.byref(step)), "get_opt_state")
        return m, v, int(step.value)

    def load_optimizer_state(self, m, v, step):
        capi.check(self._L.hx_ppo_set_opt_state_h(self._h, capi.ptr(capi.farr(m)), capi.ptr(capi.farr(v)), int(step)), "set_opt_state")

    def rng_state(self):
        a, p = capi.C.c_uint32(), capi.C.c_uint32()
        capi.check(self._L.hx_ppo_get_rng_state(self._h, capi.C.byref(a), capi.C.byref(p)), "get_rng_state")
        return int(a.value), int(p.value)

    def load_rng_state(self, act_counter, perm_counter):
        capi.check(self._L.hx_ppo_set_rng_state(self._h, int(act_counter), int(perm_counter)), "set_rng_state")

    def prof_begin(self, only=None, sample_every=1):
        """Bracket the learner's GEMM launches with HIP events on its stream (include/hx_lab.h).  only: a kernel symbol as
        returned by prof_end -- the rocprofv3 name, template arguments included -- to bracket just that symbol; sample_every = n brackets
        every n-th of those launches only (an event pair idles the stream for ~7 us on either side of the launch)."""
        capi.check(self._L.hx_ppo_prof_begin(self._h, None if only is None else only.encode(), int(sample_every)), "prof_begin")

    def prof_end(self):
        rows = (capi.ProfRow * 64)()
        n = capi.C.c_int(0)
        capi.check(self._L.hx_ppo_prof_end(self._h, rows, 64, capi.C.byref(n)), "prof_end")
        ks = [dict(name=rows[i].symbol.decode(), ms=float(rows[i].ms), launches=int(rows[i].launches), flops=float(rows[i].flops))
              for i in range(min(n.value, 64))]
        for k in ks:
            k["tflops"] = k["flops"] / (k["ms"] * 1e-3) / 1e12 if k["ms"] > 0 else 0.0
        return dict(kernels=ks)

    def close(self):
        if self._h:
            self._L.hx_ppo_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
