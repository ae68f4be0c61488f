"""OnPolicyRunner: the reference's training loop over the HIP env + learner.

Surface follows humanoid/algo/ppo/on_policy_runner.py: __init__(env, train_cfg, log_dir, device) :47-91,
learn(num_learning_iterations, init_at_random_ep_len) :93-177, log :179-276, save :278-287, load :289-295,
get_inference_policy :297-301; attributes .alg, .current_learning_iteration, .tot_timesteps.

Differences that are deliberate:
  * tensorboard / wandb sinks (absent here, and wandb.init is a network call) are replaced by a JSONL file
    with the SAME scalar names (Episode/rew_*, Loss/*, Policy/mean_noise_std, Perf/*, Train/*), so curves can
    be overlaid on a reference run's export.
  * rewards/dones never leave the device inside the rollout; episode statistics are accumulated by the env
    kernel and read once per iteration, so the loop has no per-step host synchronisation.
  * with a `comm` (isaac_amd.parallel) the gradient of every optimiser step is all-reduced over RCCL.
Checkpoints keep the reference's torch.save dict layout (model_state_dict / optimizer_state_dict / iter / infos).
"""
import json
import os
import statistics
import time
from collections import deque
from datetime import datetime

import numpy as np

from .ppo import PPO, ActorCritic

_CLASSES = {"ActorCritic": ActorCritic, "PPO": PPO}


class OnPolicyRunner:
    def __init__(self, env, train_cfg, log_dir=None, device="cuda:0", comm=None):
        self.cfg = train_cfg["runner"]
        self.alg_cfg = train_cfg["algorithm"]
        self.policy_cfg = train_cfg["policy"]
        self.all_cfg = train_cfg
        self.wandb_run_name = (datetime.now().strftime("%b%d_%H-%M-%S") + "_" + train_cfg["runner"]["experiment_name"]
                               + "_" + train_cfg["runner"]["run_name"])
        self.device = device
        self.env = env
        self.comm = comm
        num_critic_obs = env.num_privileged_obs if env.num_privileged_obs is not None else env.num_obs
        actor_critic = _CLASSES[self.cfg["policy_class_name"]](env.num_obs, num_critic_obs, env.num_actions, **self.policy_cfg)
        if comm is not None and comm.world_size > 1 and not getattr(comm, "in_library", False):
            actor_critic.load_state_dict(comm.broadcast_state(actor_critic.state_dict()))      # HxComm broadcasts on the device instead
        alg_kw = dict(self.alg_cfg)
        if "seed" in train_cfg and "seed" not in alg_kw:      # exploration noise keyed by seed + rank, permutation by seed
            alg_kw["seed"] = train_cfg["seed"]
        # The rollout's background critic yields the matrix pipes to the actor (HX_CRITIC_YIELD).  That pays while the
        # env step leaves the critic slack; on the ground plane the step is a third shorter and it does not (the critic's backlog
        # then lands behind the rollout: profiles/r03_p_grouped_launches.txt), so the runner -- which knows the terrain -- creates
        # the learner with the yield off there unless the variable is set.  Scheduling only: results are the same.
        self.alg = _CLASSES[self.cfg["algorithm_class_name"]](actor_critic, device=device, stream=getattr(env, "stream", None),
                                                              comm=comm, **alg_kw)
        plane = getattr(getattr(getattr(env, "cfg", None), "terrain", None), "mesh_type", None) == "plane"
        if plane and hasattr(self.alg, "creation_knobs"):
            self.alg.creation_knobs.setdefault("HX_CRITIC_YIELD", "0")
        self.num_steps_per_env = self.cfg["num_steps_per_env"]
        self.save_interval = self.cfg["save_interval"]
        # Rollout storage of the observations (runner.observation_storage, not a reference key): "frames" = every robot's frames
        # once (single-frame storage, include/hx_ppo.h), "rows" = the reference's stacked rows, "auto" (default) = by measurement
        # (profiles/r03_g_storage.txt, 1 x MI355X fp32): frames are 0.9 % / 1.6 % faster at 16 384 / 65 536 robots per GPU and
        # 12 x smaller; at 4096 robots the gathered first-layer loaders cost more than the stacking and gather launches they
        # replace (-3 %), so small batches keep the rows.  Results are bit-identical either way (tests/test_gpu_runner.py).
        mode = self.cfg.get("observation_storage", "auto")
        if mode not in ("auto", "frames", "rows"):
            raise ValueError("runner.observation_storage must be 'auto', 'frames' or 'rows'")
        frames = getattr(env, "frame_dims", None) if (hasattr(env, "_h") and hasattr(self.alg, "rollout")) else None
        if mode == "rows" or (mode == "auto" and env.num_envs < 16384):
            frames = None
        self.alg.init_storage(env.num_envs, self.num_steps_per_env, [env.num_obs], [env.num_privileged_obs], [env.num_actions],
                              obs_ld=getattr(env, "obs_ld", None), priv_ld=getattr(env, "priv_ld", None), frames=frames)
        # one writer per job: ranks other than 0 neither create directories nor write scalars or checkpoints (their
        # parameters are bit-identical to rank 0's); Episode/* and Train/* scalars are summed over ranks in log()
        self.is_chief = comm is None or comm.rank == 0
        self.log_dir = log_dir if self.is_chief else None
        # Episode statistics are a COLLECTIVE when data parallel (every rank must take part or none), so the decision cannot
        # depend on a per-rank argument: the common idiom `log_dir if rank == 0 else None` would leave rank 0 alone inside
        # an all-reduce.  Decided once, here, from the maximum of the flag over ranks.
        want = 1.0 if log_dir is not None else 0.0
        if comm is not None and comm.world_size > 1:
            want = comm.max_over_ranks(want)
        self.collect_stats = want > 0.0
        self.writer = None
        self.tot_timesteps = 0
        self.tot_time = 0
        self.current_learning_iteration = 0
        self.last_perf = {}
        self.iteration_times = []       # collection + learning seconds of every iteration run so far (bench.py: median beside the mean)
        self.env.reset()

    # ------------------------------------------------------------------ training loop
    def learn(self, num_learning_iterations, init_at_random_ep_len=False):
        if self.log_dir is not None and self.writer is None:
            os.makedirs(self.log_dir, exist_ok=True)
            self.writer = open(os.path.join(self.log_dir, "scalars.jsonl"), "a")
        env, alg = self.env, self.alg
        if init_at_random_ep_len:
            try:
                import torch
                r = torch.randint(0, int(env.max_episode_length), (env.num_envs,)).numpy()
            except ImportError:                          # pragma: no cover
                r = np.random.randint(0, int(env.max_episode_length), env.num_envs)
            env.episode_length_buf = r.astype(np.int32)
        shards = getattr(env, "shards", None)
        if shards is not None:
            return self._learn_pipelined(num_learning_iterations)
        obs = env.get_observations()
        privileged_obs = env.get_privileged_observations()
        critic_obs = privileged_obs if privileged_obs is not None else obs
        alg.actor_critic.train()
        rewbuffer, lenbuffer = deque(maxlen=100), deque(maxlen=100)
        tot_iter = self.current_learning_iteration + num_learning_iterations
        for it in range(self.current_learning_iteration, tot_iter):
            start = time.time()
            if hasattr(env, "_h") and hasattr(alg, "rollout"):
                alg.rollout([env], self.num_steps_per_env)          # the same 60-step loop, issued from C
                obs, privileged_obs = env.get_observations(), env.get_privileged_observations()
                critic_obs = privileged_obs if privileged_obs is not None else obs
            else:
                for _ in range(self.num_steps_per_env):
                    actions = alg.act(obs, critic_obs)
                    obs, privileged_obs, rewards, dones, infos = env.step(actions)
                    critic_obs = privileged_obs if privileged_obs is not None else obs
                    alg.process_env_step(rewards, dones, infos)
            env.sync()
            stop = time.time()
            collection_time = stop - start
            start = stop
            alg.compute_returns(critic_obs)
            mean_value_loss, mean_surrogate_loss = alg.update()      # synchronises once (loss scalars)
            stop = time.time()
            learn_time = stop - start
            self.last_perf = dict(collection_time=collection_time, learn_time=learn_time,
                                  fps=self.num_steps_per_env * env.num_envs / (collection_time + learn_time))
            self.iteration_times.append(collection_time + learn_time)
            if self.collect_stats:
                ep_info, n_ep = self._episode_stats_all_ranks()
                self.log(dict(it=it, tot_iter=tot_iter, collection_time=collection_time, learn_time=learn_time,
                              mean_value_loss=mean_value_loss, mean_surrogate_loss=mean_surrogate_loss,
                              ep_info=ep_info, n_ep=n_ep, rewbuffer=rewbuffer, lenbuffer=lenbuffer))
                if self.log_dir is not None and it % self.save_interval == 0:
                    self.save(os.path.join(self.log_dir, "model_{}.pt".format(it)))
        self.current_learning_iteration += num_learning_iterations
        if self.log_dir is not None:
            self.save(os.path.join(self.log_dir, "model_{}.pt".format(self.current_learning_iteration)))

    def _learn_pipelined(self, num_learning_iterations):
        """The same loop over a PipelinedHectorEnv: every shard advances on its own stream, so the GPU overlaps
        one shard's env-step kernel with the other shards' policy GEMMs.  Nothing here synchronises per step."""
        env, alg = self.env, self.alg
        shards = env.shards
        obs = [s.get_observations() for s in shards]
        priv = [s.get_privileged_observations() for s in shards]
        tot_iter = self.current_learning_iteration + num_learning_iterations
        for it in range(self.current_learning_iteration, tot_iter):
            start = time.time()
            alg.rollout(shards, self.num_steps_per_env)
            obs = [s.get_observations() for s in shards]
            priv = [s.get_privileged_observations() for s in shards]
            env.sync()
            stop = time.time()
            collection_time = stop - start
            start = stop
            alg.compute_returns_shards([(priv[h], sh.env_lo, sh.num_envs, sh.stream) for h, sh in enumerate(shards)])
            mean_value_loss, mean_surrogate_loss = alg.update()
            stop = time.time()
            learn_time = stop - start
            self.last_perf = dict(collection_time=collection_time, learn_time=learn_time,
                                  fps=self.num_steps_per_env * env.num_envs / (collection_time + learn_time))
            self.iteration_times.append(collection_time + learn_time)
            if self.collect_stats:
                ep_info, n_ep = self._episode_stats_all_ranks()
                self.log(dict(it=it, tot_iter=tot_iter, collection_time=collection_time, learn_time=learn_time,
                              mean_value_loss=mean_value_loss, mean_surrogate_loss=mean_surrogate_loss,
                              ep_info=ep_info, n_ep=n_ep, rewbuffer=None, lenbuffer=None))
                if self.log_dir is not None and it % self.save_interval == 0:
                    self.save(os.path.join(self.log_dir, "model_{}.pt".format(it)))
        self.current_learning_iteration += num_learning_iterations
        if self.log_dir is not None:
            self.save(os.path.join(self.log_dir, "model_{}.pt".format(self.current_learning_iteration)))

    def _episode_stats_all_ranks(self):
        """Episode/* and Train/* scalars over the envs of ALL ranks.  A collective when world_size > 1, so every rank calls
        it: ONE all-reduce of a packed vector [n_ep | n_ep * value ...], i.e. the means are weighted by the number of
        episodes each rank finished in this iteration (a rank without finished episodes contributes nothing instead of a
        stale value)."""
        info, n_ep = self.env.episode_stats()
        self._train_stats = (self.env.last_episode_return, self.env.last_episode_length)
        if self.comm is not None and self.comm.world_size > 1:
            keys = sorted(info)
            w = float(n_ep)
            vec = np.array([w, 1.0] + [w * float(info[k]) for k in keys] + [float(info[k]) for k in keys]
                           + [w * float(v) for v in self._train_stats] + [float(v) for v in self._train_stats], np.float64)
            tot = self.comm.sum_array_over_ranks(vec)
            n_all, ranks, nk = tot[0], tot[1], len(keys)
            wsum, usum = tot[2:2 + nk], tot[2 + nk:2 + 2 * nk]
            tw, tu = tot[2 + 2 * nk:4 + 2 * nk], tot[4 + 2 * nk:6 + 2 * nk]
            if n_all > 0:
                # terrain_level is a mean over ALL robots (legged_robot.py:203-204), not over finished episodes
                info = {k: float(usum[i] / ranks if k == "terrain_level" else wsum[i] / n_all) for i, k in enumerate(keys)}
                self._train_stats = tuple(float(x / n_all) for x in tw)
            else:                       # nobody finished an episode: plain mean of the (persisting) per-rank values
                info = {k: float(usum[i] / ranks) for i, k in enumerate(keys)}
                self._train_stats = tuple(float(x / ranks) for x in tu)
            n_ep = int(round(n_all))
        return info, n_ep

    def log(self, locs, width=80, pad=35):
        world = 1 if self.comm is None else self.comm.world_size
        self.tot_timesteps += self.num_steps_per_env * self.env.num_envs * world
        self.tot_time += locs["collection_time"] + locs["learn_time"]
        iteration_time = locs["collection_time"] + locs["learn_time"]
        fps = int(self.num_steps_per_env * self.env.num_envs * world / iteration_time)
        scalars = {}
        for k, v in locs["ep_info"].items():
            scalars["Episode/" + k] = v
        mean_std = float(np.mean(self.alg.actor_critic.std))
        scalars.update({"Loss/value_function": locs["mean_value_loss"], "Loss/surrogate": locs["mean_surrogate_loss"],
                        "Loss/learning_rate": self.alg.learning_rate, "Policy/mean_noise_std": mean_std,
                        "Perf/total_fps": fps, "Perf/collection time": locs["collection_time"],
                        "Perf/learning_time": locs["learn_time"]})
        if locs["n_ep"] > 0:
            scalars["Train/mean_reward"], scalars["Train/mean_episode_length"] = self._train_stats
        if self.writer is not None and self.is_chief:
            self.writer.write(json.dumps({"it": locs["it"], "tot_timesteps": self.tot_timesteps, "tot_time": self.tot_time, **scalars}) + "\n")
            self.writer.flush()
        if self.comm is not None and self.comm.rank != 0:
            return
        head = f" \033[1m Learning iteration {locs['it']}/{self.current_learning_iteration + locs['tot_iter'] - self.current_learning_iteration} \033[0m "
        lines = [f"{'#' * width}", f"{head.center(width, ' ')}", "",
                 f"{'Computation:':>{pad}} {fps:.0f} steps/s (collection: {locs['collection_time']:.3f}s, learning {locs['learn_time']:.3f}s)",
                 f"{'Value function loss:':>{pad}} {locs['mean_value_loss']:.4f}",
                 f"{'Surrogate loss:':>{pad}} {locs['mean_surrogate_loss']:.4f}",
                 f"{'Mean action noise std:':>{pad}} {mean_std:.2f}"]
        if locs["n_ep"] > 0:
            lines += [f"{'Mean reward:':>{pad}} {self._train_stats[0]:.2f}",
                      f"{'Mean episode length:':>{pad}} {self._train_stats[1]:.2f}"]
        for k, v in locs["ep_info"].items():
            lines.append(f"{'Mean episode ' + k + ':':>{pad}} {v:.4f}")
        lines += ["-" * width, f"{'Total timesteps:':>{pad}} {self.tot_timesteps}",
                  f"{'Iteration time:':>{pad}} {iteration_time:.2f}s", f"{'Total time:':>{pad}} {self.tot_time:.2f}s"]
        print("\n".join(lines))

    # ------------------------------------------------------------------ checkpoints (torch is used ONLY here)
    def _opt_state_dict(self):
        m, v, step = self.alg.optimizer_state()
        shapes = self.alg.actor_critic.tensor_shapes()
        state, o = {}, 0
        import torch
        for i, (name, shape) in enumerate(shapes.items()):
            n = int(np.prod(shape))
            state[i] = {"step": torch.tensor(float(step)), "exp_avg": torch.from_numpy(m[o:o + n].reshape(shape).copy()),
                        "exp_avg_sq": torch.from_numpy(v[o:o + n].reshape(shape).copy())}
            o += n
        groups = [{"lr": self.alg.learning_rate, "betas": (0.9, 0.999), "eps": 1e-08, "weight_decay": 0, "amsgrad": False,
                   "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                   "params": list(range(len(shapes)))}]
        return {"state": state, "param_groups": groups}

    def save(self, path, infos=None):
        import torch
        sd = {k: torch.from_numpy(v) for k, v in self.alg.actor_critic.state_dict().items()}
        if not self.is_chief:
            return
        # the reference's four keys (on_policy_runner.py:278-287) + the positions of the learner's counter-based random
        # streams, so that a resumed run does not replay the exploration noise and permutations from the start
        torch.save({"model_state_dict": sd, "optimizer_state_dict": self._opt_state_dict(),
                    "iter": self.current_learning_iteration, "infos": infos,
                    "hx_rng_state": list(self.alg.rng_state())}, path)

    def load(self, path, load_optimizer=True):
        import torch
        loaded = torch.load(path, map_location="cpu", weights_only=False)
        self.alg.actor_critic.load_state_dict(loaded["model_state_dict"])
        if load_optimizer and loaded.get("optimizer_state_dict"):
            st = loaded["optimizer_state_dict"]["state"]
            if st:
                m = np.concatenate([st[i]["exp_avg"].numpy().reshape(-1) for i in sorted(st)])
                v = np.concatenate([st[i]["exp_avg_sq"].numpy().reshape(-1) for i in sorted(st)])
                self.alg.load_optimizer_state(m, v, int(float(st[0]["step"])))
            self.alg.learning_rate = loaded["optimizer_state_dict"]["param_groups"][0]["lr"]
        if loaded.get("hx_rng_state") is not None:
            self.alg.load_rng_state(*loaded["hx_rng_state"])
        self.current_learning_iteration = loaded["iter"]
        return loaded["infos"]

    def get_inference_policy(self, device=None):
        self.alg.actor_critic.eval()
        return self.alg.actor_critic.act_inference
