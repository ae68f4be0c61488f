"""ORACLE-SIDE TOOL (test infrastructure, never shipped): Python handle of the HOST build of the simulator's single-source
device code (oracle/host/hx_host.cpp -> oracle/_host/libhx_host.so, built by oracle/host/Makefile).

Used by tests/ (the kernels' own text against the numpy oracle on the CPU, and under AddressSanitizer / UBSan), by
bench.py's cpu_baseline leg and by tools/ parameter studies.  The product never imports this package."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(os.path.dirname(_HERE), "_host")
_libs = {}


def build(asan=False):
    """Serialised across processes (flock): every rank of a job started by a launcher may call this at the same time.
    asan: False (the optimised build), True (AddressSanitizer + UBSan), "nogap2" (optimised, -DHX_NO_GAP2)."""
    import fcntl
    target = "nogap2" if asan == "nogap2" else "asan" if asan else "all"
    os.makedirs(OUT_DIR, exist_ok=True)
    with open(os.path.join(OUT_DIR, ".build.lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            subprocess.check_call(["make", "-s", "-C", _HERE, target])
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)
    return os.path.join(OUT_DIR, _so_name(asan))


def _so_name(asan):
    return "libhx_host_nogap2.so" if asan == "nogap2" else "libhx_host_asan.so" if asan else "libhx_host.so"


def lib(asan=False, build_if_missing=True):
    key = asan if asan == "nogap2" else bool(asan)
    if key in _libs:
        return _libs[key]
    path = os.path.join(OUT_DIR, _so_name(asan))
    if build_if_missing and os.path.exists(os.path.join(_HERE, "Makefile")):
        build(asan)          # make is a no-op when the binary is current; a failed build is never masked by a stale binary
    L = C.CDLL(path)
    vp = C.c_void_p
    L.hxh_create.restype = vp
    L.hxh_create.argtypes = [vp, vp, vp, vp, vp, C.c_uint64]
    L.hxh_destroy.argtypes = [vp]
    L.hxh_set_terrain.argtypes = [vp, vp, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float]
    L.hxh_reset_all.argtypes = [vp, vp]
    L.hxh_step.argtypes = [vp, vp, vp]
    L.hxh_buffer.restype = vp
    L.hxh_buffer.argtypes = [vp, C.c_int]
    L.hxh_state.restype = vp
    L.hxh_state.argtypes = [vp]
    L.hxh_state_size.argtypes = [vp]
    L.hxh_get_state.argtypes = [vp, vp, vp, vp]
    L.hxh_set_state.argtypes = [vp, vp, vp, vp]
    L.hxh_set_episode_length.argtypes = [vp, vp]
    L.hxh_set_step_counter.argtypes = [vp, C.c_int64]
    L.hxh_set_commands.argtypes = [vp, vp]
    _libs[key] = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class HostEnv:
    """The env step of isaac_amd.envs.hector_env.HectorFreeEnv / HectorFullFreeEnv served by the host build.  Takes the
    product's own host-side derivation of the flat config (HectorFreeEnv._derive: no library call), so both see the
    same numbers."""

    def __init__(self, cfg, creation=None, init_pack=None, full=False, asan=False, task=None):
        from isaac_amd import capi
        from isaac_amd.envs.hector_env import HectorFreeEnv, HectorFullFreeEnv, XBotLFreeEnv
        cls = {"hector": HectorFreeEnv, "hector_full": HectorFullFreeEnv, "humanoid_ppo": XBotLFreeEnv}[task or ("hector_full" if full else "hector")]
        self.host = cls.__new__(cls)
        self.host.cfg = cfg
        c, friction, mass, start, terrain_grid, rough = self.host._derive(cfg, None, creation, None)
        self.ccfg = c
        self.L = lib(asan)
        self.n, self.nd = c.num_envs, c.num_dof
        self.obs_f, self.priv_f, self.priv_stack = 11 + 3 * self.nd, cls.PRIV_BASE + 3 * self.nd, cls.PRIV_STACK
        self.obs_ld, self.priv_ld = -(-15 * self.obs_f // 4) * 4, -(-self.priv_stack * self.priv_f // 4) * 4
        seed = (int(getattr(cfg, "seed", 0)) & 0xFFFFFFFF) | (0x5EED << 32)
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        self._keep = [f32(friction), f32(mass), f32(self.host.env_origins), f32(start)]
        self.h = self.L.hxh_create(C.byref(c), *[_p(a) for a in self._keep], seed)
        assert self.h, "hxh_create failed"
        if rough:
            hts = np.ascontiguousarray(terrain_grid["heights"], np.int16)
            b = -float(terrain_grid["border_size"])
            rc = self.L.hxh_set_terrain(self.h, _p(hts), hts.shape[0], hts.shape[1], float(terrain_grid["horizontal_scale"]),
                                        float(terrain_grid["vertical_scale"]), b, b, float(self.host._wall))
            assert rc == 0
        self.capi = capi
        pk = None if init_pack is None else np.ascontiguousarray(init_pack, np.float32)
        self.L.hxh_reset_all(self.h, _p(pk))

    def _view(self, which, shape, dtype=np.float32):
        ptr = self.L.hxh_buffer(self.h, which)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dtype))), shape)

    @property
    def obs_buf(self):
        return self._view(self.capi.BUF_OBS, (self.n, self.obs_ld))[:, :15 * self.obs_f].copy()

    @property
    def privileged_obs_buf(self):
        return self._view(self.capi.BUF_PRIV, (self.n, self.priv_ld))[:, :self.priv_stack * self.priv_f].copy()

    def step(self, actions, pack=None):
        a = np.ascontiguousarray(actions, np.float32)
        pk = None if pack is None else np.ascontiguousarray(pack, np.float32)
        self.L.hxh_step(self.h, _p(a), _p(pk))
        return (self.obs_buf, self.privileged_obs_buf, self._view(self.capi.BUF_REW, (self.n,)).copy(),
                self._view(self.capi.BUF_RESET, (self.n,), np.uint8).copy().astype(bool))

    @property
    def time_out_buf(self):
        return self._view(self.capi.BUF_TIMEOUT, (self.n,), np.uint8).copy().astype(bool)

    @property
    def time_outs_visible(self):
        return self._view(self.capi.BUF_TIMEOUT_VISIBLE, (self.n,), np.uint8).copy().astype(bool)

    @property
    def torques(self):
        return self._view(self.capi.BUF_TORQUES, (self.nd, self.n)).T.copy()

    @property
    def contact_forces(self):
        return self._view(self.capi.BUF_CONTACT, (1 + self.nd, 3, self.n)).transpose(2, 0, 1).copy()

    @property
    def episode_length_buf(self):
        return self._view(self.capi.BUF_EP_LEN, (self.n,), np.int32).copy()

    @episode_length_buf.setter
    def episode_length_buf(self, v):
        a = np.ascontiguousarray(v, np.int32)
        self.L.hxh_set_episode_length(self.h, _p(a))

    @property
    def commands(self):
        return self._view(self.capi.BUF_COMMANDS, (4, self.n)).T.copy()

    @commands.setter
    def commands(self, v):
        a = np.ascontiguousarray(np.asarray(v, np.float32).reshape(self.n, 4))
        self.L.hxh_set_commands(self.h, _p(a))

    def get_state(self):
        root, q, qd = np.empty((self.n, 13), np.float32), np.empty((self.n, self.nd), np.float32), np.empty((self.n, self.nd), np.float32)
        self.L.hxh_get_state(self.h, _p(root), _p(q), _p(qd))
        return root, q, qd

    def set_state(self, root, q, qd):
        a, b, c = (np.ascontiguousarray(x, np.float32) for x in (root, q, qd))
        self.L.hxh_set_state(self.h, _p(a), _p(b), _p(c))

    def set_step_counter(self, c):
        self.L.hxh_set_step_counter(self.h, int(c))

    def base_lin_vel(self):
        st = np.ctypeslib.as_array(C.cast(self.L.hxh_state(self.h), C.POINTER(C.c_float)), (self.L.hxh_state_size(self.h), self.n))
        o = 41 + 6 * self.nd
        return st[o:o + 3].T.copy(), st[o + 3:o + 6].T.copy()

    def close(self):
        if self.h:
            self.L.hxh_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
