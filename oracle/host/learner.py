"""ORACLE-SIDE TOOL (test infrastructure, never shipped): the numpy oracle's MLP served by the compiled host restatement of the
learner's dense arithmetic (oracle/host/hx_learner_host.cpp -> oracle/_host/libhx_learner_host.so).

HostMLP is a drop-in for oracle.ppo.MLP (same forward / backward contract), HostPPOOracle a PPOOracle whose optimiser step
runs in the same library; loss head, GAE and the learning-rate schedule stay the numpy oracle's.  Used by bench.py's
cpu_baseline leg and tests/test_host_learner.py; the product never imports this package."""
import ctypes as C
import os

import numpy as np

from oracle.ppo import MLP, ActorCriticOracle, PPOOracle
from . import OUT_DIR, build

F = np.float32
_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(os.path.join(OUT_DIR, "libhx_learner_host.so"))
        vp, i, f = C.c_void_p, C.c_int, C.c_float
        L.hxl_linear_forward.argtypes = [i, i, i, vp, vp, vp, vp, i]
        L.hxl_linear_wgrad.argtypes = [i, i, i, vp, vp, vp, vp]
        L.hxl_linear_dgrad.argtypes = [i, i, i, vp, vp, vp, vp]
        L.hxl_sumsq.argtypes = [vp, C.c_int64]
        L.hxl_sumsq.restype = C.c_double
        L.hxl_adam.argtypes = [vp, vp, vp, vp, C.c_int64, f, f, f]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class HostMLP(MLP):
    """Outputs live in buffers that are reused from call to call (keyed by role and shape): a fresh multi-MB numpy array per
    layer per call is an mmap + page faults every time, which costs more than the GEMM on some hosts.  A result is therefore
    valid until the next forward / backward of the same network with the same row count."""

    def _buf(self, key, shape):
        cache = self.__dict__.setdefault("_cache", {})
        b = cache.get((key, shape))
        if b is None:
            b = cache[(key, shape)] = np.empty(shape, F)
        return b

    def forward(self, x, keep=False, bf16=False):
        assert not bf16, "the compiled host learner is fp32 (the metric's configuration)"
        L = lib()
        h = np.ascontiguousarray(x, F)
        hs = [h]
        n = len(self.W)
        for i in range(n):
            out = self._buf(("h", i), (h.shape[0], self.W[i].shape[0]))
            L.hxl_linear_forward(h.shape[0], self.W[i].shape[0], self.W[i].shape[1], _p(h), _p(self.W[i]), _p(self.b[i]), _p(out), int(i < n - 1))
            h = out
            hs.append(h)
        return (h, hs) if keep else h

    def backward(self, hs, dout, bf16=False):
        assert not bf16
        L = lib()
        n = len(self.W)
        dW, db = [None] * n, [None] * n
        dz = np.ascontiguousarray(dout, F)
        for i in range(n - 1, -1, -1):
            M, N, K = dz.shape[0], self.W[i].shape[0], self.W[i].shape[1]
            dW[i], db[i] = self._buf(("dW", i), (N, K)), self._buf(("db", i), (N,))
            L.hxl_linear_wgrad(M, N, K, _p(dz), _p(np.ascontiguousarray(hs[i], F)), _p(dW[i]), _p(db[i]))
            if i > 0:
                nxt = self._buf(("dz", i), (M, K))
                L.hxl_linear_dgrad(M, N, K, _p(dz), _p(self.W[i]), _p(np.ascontiguousarray(hs[i], F)), _p(nxt))
                dz = nxt
        return dW, db


def host_actor_critic(ac):
    """the same parameters (shared arrays) behind compiled forward / backward"""
    out = ActorCriticOracle.__new__(ActorCriticOracle)
    out.actor, out.critic, out.std = HostMLP.__new__(HostMLP), HostMLP.__new__(HostMLP), ac.std
    out.actor.W, out.actor.b, out.critic.W, out.critic.b = ac.actor.W, ac.actor.b, ac.critic.W, ac.critic.b
    for net in (out.actor, out.critic):
        net.W = [np.ascontiguousarray(w, F) for w in net.W]
        net.b = [np.ascontiguousarray(b, F) for b in net.b]
    return out


class HostPPOOracle(PPOOracle):
    """PPOOracle on the compiled host learner: ac must come from host_actor_critic()."""

    def optimizer_step(self, grads):
        L = lib()
        grads = [np.ascontiguousarray(g, F) for g in grads]
        total = float(np.sqrt(sum(L.hxl_sumsq(_p(g), g.size) for g in grads)))
        coef = min(1.0, self.max_grad_norm / (total + 1e-6))
        self.t += 1
        bc1, bc2 = 1 - 0.9 ** self.t, 1 - 0.999 ** self.t
        for p, g, m, v in zip(self.ac.params(), grads, self.m, self.v):
            assert p.flags.c_contiguous and m.flags.c_contiguous and v.flags.c_contiguous
            L.hxl_adam(_p(p), _p(g), _p(m), _p(v), p.size, F(coef), F(self.lr / bc1), F(np.sqrt(bc2)))
        return total
