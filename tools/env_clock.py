#!/usr/bin/env python3
"""GPU: the shader clock the env-step kernel runs at, from in-kernel stamps (d s_memtime / d s_memrealtime x 100 MHz, summed over
all waves; -DHX_STEP_PROF build): (a) alone, back-to-back launches; (b) inside the rollout, i.e. beside the fused actor's launches
and the background critic's persistent MFMA waves.  Restores the normal build afterwards.  usage: python tools/env_clock.py [envs]"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("HX_STEP_PROF_CHILD") != "1":
    env = dict(os.environ, HX_EXTRA_FLAGS_HX_SIM="-DHX_STEP_PROF", HX_STEP_PROF_CHILD="1")
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
L = capi.lib()
waves = (n + 7) // 8


def read(label, launches):
    out = np.zeros(18, np.int64)
    capi.check(L.hx_sim_prof(env._h, 0, out.ctypes.data), "prof")
    print(f"{label}: in-kernel clock {0.1 * out[16] / out[17]:.3f} GHz; a wave lives {out[17] / (waves * launches) / 100.0:.1f} us and {out[16] / (waves * launches) / 1e3:.1f} k cycles on average ({launches} launches)")


act = capi.DeviceBuffer.from_host((0.3 * np.random.default_rng(0).standard_normal((n, env.num_actions))).astype(np.float32))
for _ in range(50):
    L.hx_sim_step(env._h, act.ptr, None)
capi.check(L.hx_sim_prof(env._h, 1, None), "prof")
for _ in range(300):
    L.hx_sim_step(env._h, act.ptr, None)
read("alone (300 back-to-back env steps, stacking kernels between them)", 300)
wv = np.zeros(waves, np.int64)
capi.check(L.hx_sim_prof_waves(env._h, wv.ctypes.data, waves), "prof_waves")
us = wv / 300 / 100.0
q = np.percentile(us, [0, 10, 25, 50, 75, 90, 99, 100])
print("per-wave lifetime over the 300 launches, us: min %.0f p10 %.0f p25 %.0f median %.0f p75 %.0f p90 %.0f p99 %.0f max %.0f; mean %.1f" % (*q, us.mean()))
cols = getattr(env, "terrain_types", None)
per_type = waves // 20
print("by terrain column (20 columns of %d waves, env order = column-major as in _get_env_origins): " % per_type + " ".join("%.0f" % us[c * per_type:(c + 1) * per_type].mean() for c in range(20)))
print("slowest wave per column: " + " ".join("%.0f" % us[c * per_type:(c + 1) * per_type].max() for c in range(20)))
runner = OnPolicyRunner(env, class_to_dict(HectorCfgPPO()), log_dir=None, device="cuda:0")
runner.learn(2, init_at_random_ep_len=True)
capi.check(L.hx_sim_prof(env._h, 1, None), "prof")
runner.learn(3, init_at_random_ep_len=False)
env.sync()
read("inside the rollout (3 iterations: beside the actor's launches and the background critic)", 3 * runner.num_steps_per_env)
