#!/usr/bin/env python3
"""GPU: when the waves of ONE env-step launch start and end, and on which SIMDs they run (-DHX_STEP_PROF build; stamps of the
device-wide 100 MHz counter, include/hx_lab.h hx_sim_prof_last): (a) alone, back-to-back launches; (b) the last env step of a rollout
(beside a background-critic batch).  Restores the normal build afterwards.  usage: python tools/env_waves.py [envs]"""
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
from collections import Counter
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


def last(label):
    out = np.zeros((18, waves), np.int64)
    capi.check(L.hx_sim_prof_last(env._h, out.ctypes.data, waves), "prof_last")
    st, en, hw = out[0] / 100.0, out[1] / 100.0, out[2]
    t0 = st.min()
    so, eo, life = st - t0, en - t0, en - st
    # HW_ID (gfx9 layout): wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13; XCC_ID in the high word
    cu = ((hw >> 32) << 16) | ((hw & 0xffffffff) >> 8 & 0xff)
    simd = (cu << 2) | ((hw >> 4) & 3)
    per_cu = Counter(Counter(cu.tolist()).values())
    per_simd = Counter(Counter(simd.tolist()).values())
    q = lambda x: "p50 %.1f p90 %.1f p99 %.1f max %.1f" % tuple(np.percentile(x, [50, 90, 99, 100]))
    print(f"{label}: span (last end - first start) {eo.max():.1f} us")
    print(f"   start after the first wave's, us: {q(so)}     lifetime, us: mean {life.mean():.1f} {q(life)}")
    print(f"   CUs used {len(set(cu.tolist()))}, waves per CU {dict(sorted(per_cu.items()))}, waves per SIMD {dict(sorted(per_simd.items()))}")
    late = np.argsort(eo)[-8:][::-1]
    print("   the 8 waves that end last: " + "; ".join("start +%.0f, lives %.0f" % (so[i], life[i]) for i in late))
    order = np.argsort(so)
    print("   lifetime by start rank: " + ", ".join("waves %d-%d: mean %.0f max %.0f" % (a, b - 1, life[order[a:b]].mean(), life[order[a:b]].max())
                                                   for a, b in ((0, 8), (8, 32), (32, 128), (128, 256), (256, waves))))
    print("   the 8 that end last: block " + " ".join("%d(xcc %d, hw %#x)" % (i, hw[i] >> 32, hw[i] & 0xffffffff) for i in late))
    print("   lifetime by block index: " + ", ".join("blocks %d-%d: mean %.0f max %.0f" % (a, b - 1, life[a:b].mean(), life[a:b].max())
                                                    for a, b in ((0, 8), (8, 32), (32, 128), (128, 256), (256, waves))))
    ph = out[3:12].astype(np.float64) / 1e3        # k cycles per phase
    names = ["fetch", "action", "kinem", "contact", "inertia", "solve", "accel", "guard", "glue"]
    slow = np.argsort(life)[-32:]; rest = np.argsort(life)[:-32]
    print("   phase k-cycles, the 32 slowest waves / the others: " + ", ".join("%s %.1f / %.1f" % (nm, ph[k][slow].mean(), ph[k][rest].mean()) for k, nm in enumerate(names))
          + "; sum %.0f / %.0f" % (ph[:, slow].sum(0).mean(), ph[:, rest].sum(0).mean()))
    vis = out[12:18].astype(np.float64)
    print("   contact-loop visits per env step (wave-level), the 32 slowest / the others: %.1f / %.1f; of them with a contact %.1f / %.1f; sides that asked, per visit %.2f / %.2f; visits asked for by ONE side %.1f / %.1f; point groups past the per-point pooled test %.1f / %.1f"
          % (vis[0][slow].mean(), vis[0][rest].mean(), vis[1][slow].mean(), vis[1][rest].mean(), vis[2][slow].sum() / max(vis[0][slow].sum(), 1), vis[2][rest].sum() / max(vis[0][rest].sum(), 1),
             vis[3][slow].mean(), vis[3][rest].mean(), vis[4][slow].mean(), vis[4][rest].mean()))
    print("   contact k-cycles per visit: slowest %.1f, others %.1f;  correlation of a wave's lifetime with its visits %.2f, with its start offset %.2f"
          % (ph[3][slow].sum() / max(vis[0][slow].sum(), 1), ph[3][rest].sum() / max(vis[0][rest].sum(), 1), np.corrcoef(life, vis[0])[0, 1], np.corrcoef(life, so)[0, 1]))
    second = so > 20.0
    print(f"   waves that start more than 20 us after the first: {int(second.sum())} (lifetime mean {life[second].mean() if second.any() else 0:.1f} us)")


act = capi.DeviceBuffer.from_host((0.3 * np.random.default_rng(0).standard_normal((n, env.num_actions))).astype(np.float32))
for _ in range(50):
    L.hx_sim_step(env._h, act.ptr, None)
capi.check(L.hx_sim_prof(env._h, 1, None), "prof")
for k in range(3):
    for _ in range(7):
        L.hx_sim_step(env._h, act.ptr, None)
    last("alone, back-to-back launches, sample %d" % k)
if os.environ.get("ENV_WAVES_ALONE_ONLY") == "1":
    sys.exit(0)
runner = OnPolicyRunner(env, class_to_dict(HectorCfgPPO()), log_dir=None, device="cuda:0")
runner.learn(2, init_at_random_ep_len=True)
for k in range(5):
    runner.learn(1, init_at_random_ep_len=False)
    env.sync()
    last("rollout, last env step of iteration %d (beside a critic batch)" % k)
# (a') the same without the stacking launch between the env steps (hx_sim_step_frames into a scratch frame slot; only the 16-workgroup
# bookkeeping kernel runs in between): does an L2 that still holds the kernel's code and constants change who is slow?
import ctypes as C


class FrameSlot(C.Structure):
    _fields_ = [("obs", C.c_void_p), ("obs_env_stride", C.c_int64), ("priv", C.c_void_p), ("priv_env_stride", C.c_int64),
                ("obs_kz", C.c_void_p), ("priv_kz", C.c_void_p)]


fo, fp = env.frame_dims[0], env.frame_dims[1] if hasattr(env, "frame_dims") else (41, 70)
bo, bp = capi.DeviceBuffer(n * 64 * 4), capi.DeviceBuffer(n * 128 * 4)
kz1, kz2 = capi.DeviceBuffer(n * 4), capi.DeviceBuffer(n * 4)
slot = FrameSlot(bo.ptr, 64, bp.ptr, 128, kz1.ptr, kz2.ptr)
L.hx_sim_step_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
for k in range(3):
    for _ in range(7):
        capi.check(L.hx_sim_step_frames(env._h, act.ptr, None, C.byref(slot), None, None, None), "step_frames")
    last("alone, NO stacking launch between the env steps, sample %d" % k)
