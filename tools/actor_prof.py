#!/usr/bin/env python3
"""GPU: where the fused rollout actor's time goes.  Rebuilds libhx.so with -DHX_ACTOR_PROF (thread 0 of every workgroup stamps the
100 MHz wall clock at phase boundaries), runs a few training iterations and prints, over the workgroups of the LAST actor launch, the
mean duration of each phase and the spread of start / end times; then restores the normal build.  usage: python tools/actor_prof.py [envs]"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("ISAAC_ACTOR_PROF_CHILD") != "1":
    env = dict(os.environ, HX_EXTRA_FLAGS_HX_PPO="-DHX_ACTOR_PROF", ISAAC_ACTOR_PROF_CHILD="1")
    subprocess.check_call([sys.executable, "-c", "from isaac_amd import build; build.build(force=True)"], env=env, cwd=ROOT)
    rc = subprocess.call([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, cwd=ROOT)
    subprocess.check_call([sys.executable, "-c", "from isaac_amd import build; build.build(force=True)"], cwd=ROOT)
    sys.exit(rc)
import numpy as np
from isaac_amd import capi
from isaac_amd.envs.configs import HectorCfg, HectorCfgPPO
from isaac_amd.envs.hector_env import HectorFreeEnv, class_to_dict
from isaac_amd.algo.on_policy_runner import OnPolicyRunner
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = HectorCfg(); cfg.env.num_envs = n; cfg.seed = 5
np.random.seed(5)
env = HectorFreeEnv(cfg)
runner = OnPolicyRunner(env, class_to_dict(HectorCfgPPO()), log_dir=None, device="cuda:0")
runner.learn(3, init_at_random_ep_len=True)
env.sync()
blocks = (n + 15) // 16
st = np.zeros((blocks, 8), np.int64)
capi.check(capi.lib().hx_ppo_actor_stamps(runner.alg._h, st.ctypes.data, blocks), "stamps")
t0 = st[:, 0].min()
us = (st[:, :7] - t0) / 100.0
names = ["rows staged", "layer 1 (616 -> 512)", "layer 2 (512 -> 256)", "layer 3 (256 -> 128)", "head + sampling", "log-prob"]
print(f"fused actor, {blocks} workgroups, last launch of the rollout: first workgroup starts at 0; starts spread over {us[:, 0].max():.1f} us, last workgroup ends at {us[:, 6].max():.1f} us")
for i, nm in enumerate(names):
    d = us[:, i + 1] - us[:, i]
    print(f"  {nm:24s} mean {d.mean():6.1f} us   min {d.min():6.1f}   max {d.max():6.1f}")
print(f"  {'workgroup lifetime':24s} mean {(us[:, 6] - us[:, 0]).mean():6.1f} us   min {(us[:, 6] - us[:, 0]).min():6.1f}   max {(us[:, 6] - us[:, 0]).max():6.1f}")
