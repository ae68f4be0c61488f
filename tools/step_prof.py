#!/usr/bin/env python3
"""GPU: where the env-step kernel's time goes, phase by phase.  Rebuilds libhx.so with -DHX_STEP_PROF (lane 0 of every wave
reads the shader clock at phase boundaries), runs `steps` env steps and prints the mean cycles per wave and launch of each
phase; then restores the normal build.  usage: python tools/step_prof.py [envs] [steps] [mesh] [task]"""
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
from isaac_amd.envs.configs import HectorCfg, HectorFullCfg
from isaac_amd.envs.hector_env import HectorFreeEnv, HectorFullFreeEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
mesh = sys.argv[3] if len(sys.argv) > 3 else "trimesh"
task = sys.argv[4] if len(sys.argv) > 4 else "hector"
cfg = (HectorFullCfg if task == "hector_full" else HectorCfg)(); cfg.env.num_envs = n; cfg.seed = 5
cfg.terrain.mesh_type = "plane" if mesh == "plane" else ("heightfield" if mesh == "heightfield" else "trimesh")
if mesh == "flatgrid":
    cfg.terrain.terrain_proportions = [1.0, 0, 0, 0, 0, 0, 0]
np.random.seed(5)
env = (HectorFullFreeEnv if task == "hector_full" else HectorFreeEnv)(cfg)
act = capi.DeviceBuffer.from_host((0.3 * np.random.default_rng(0).standard_normal((n, env.num_actions))).astype(np.float32))
L = capi.lib()
for _ in range(20):
    L.hx_sim_step(env._h, act.ptr, None)
capi.check(L.hx_sim_prof(env._h, 1, None), "prof")
for _ in range(steps):
    L.hx_sim_step(env._h, act.ptr, None)
out = np.zeros(18, np.int64)
capi.check(L.hx_sim_prof(env._h, 0, out.ctypes.data), "prof")
waves = (n + 7) // 8
names = ["window fetch + pooling", "action processing", "kinematics (x10)", "contact phase (x10)", "articulated inertias (x10)",
         "exchange + base solve (x10)", "accelerations, forces, integration (x10)", "guard + gather", "glue"]
per = out[:9] / (waves * steps)
visits = out[9:15] / (waves * steps * 2 * 10)       # lane 0 of each wave counts once per side_up; 10 substeps
print(f"N={n} {task} {mesh}: cycles per wave and env step (shader clock), {steps} steps, timers themselves included")
for nm, c in zip(names, per):
    print(f"  {nm:45s} {c:10.0f} cycles  {100 * c / per.sum():5.1f} %")
print(f"  {'total':45s} {per.sum():10.0f} cycles")
print("  contact-loop shapes visited per substep (wave-level: any of the wave's 16 sides): %.2f; of them with a contact %.2f; sides that asked, per visit %.2f of 16; visits that ONE side asked for %.2f"
      % (visits[0] * 2, visits[1] * 2, visits[2] / max(visits[0], 1e-9), visits[3] * 2))
if out[17] > 0:
    print("  in-kernel clock (sum over waves of d s_memtime / d s_memrealtime x 100 MHz): %.3f GHz; a wave lives %.1f us on average" % (0.1 * out[16] / out[17], out[17] / (waves * steps) / 100.0))
