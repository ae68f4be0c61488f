#!/usr/bin/env python3
"""Wall-clock per launch of the rollout's three kernels when each is issued back to back on its own (GPU box only):
the difference to their rocprofv3 durations is what a launch boundary costs them.
usage: step_parts.py [envs] [repeats]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HX_CRITIC_CHUNK", "1000000")          # no critic work in between
import numpy as np
from isaac_amd import capi
from isaac_amd.envs.configs import HectorCfg
from isaac_amd.envs.hector_env import HectorFreeEnv
from isaac_amd.algo.ppo import PPO, ActorCritic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
cfg = HectorCfg(); cfg.env.num_envs = n
env = HectorFreeEnv(cfg, sim_device="cuda:0", headless=True)
ac = ActorCritic(615, 1050, 10, [512, 256, 128], [768, 256, 128])
alg = PPO(ac, num_learning_epochs=2, num_mini_batches=4, stream=env.stream)
alg.init_storage(n, 60, [615], [1050], [10], obs_ld=env.obs_ld, priv_ld=env.priv_ld)
L = capi.lib()
obs, priv = env.get_observations(), env.get_privileged_observations()
act = alg.act(obs, priv)
for _ in range(20):
    env.step(act)
env.sync()

def timed(label, fn):
    for _ in range(10): fn()
    env.sync(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    env.sync(); dt = (time.perf_counter() - t0) / reps
    print(f"{label:58s} {1e6 * dt:8.1f} us per call")

out = capi.C.c_void_p()
# slot 0 of the learner's storage holds the rows after the first act(); reading them in place avoids the hand-over copy
obs_p, priv_p = alg.buffer(capi.PPO_BUF_OBS, (1,)).ptr, alg.buffer(capi.PPO_BUF_PRIV, (1,)).ptr
timed("hx_ppo_act (fused actor kernel alone, slot 0)", lambda: L.hx_ppo_act(alg._h, obs_p, priv_p, None, capi.C.byref(out)))
timed("hx_sim_step (env-step kernel + stack kernel)", lambda: L.hx_sim_step(env._h, act.ptr, None))
def both():
    L.hx_ppo_act(alg._h, obs_p, priv_p, None, capi.C.byref(out)); L.hx_sim_step(env._h, act.ptr, None)
timed("hx_ppo_act + hx_sim_step (the three launches of a rollout step)", both)
