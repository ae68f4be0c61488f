"""Data parallelism for the hector hot path: one process per GPU, envs sharded, ONE collective per optimiser step.

The reference is single-process (an unused --horovod flag is its only trace of DP).  The path
shards naturally -- environments never interact -- so rank r owns its own simulator + rollout storage and the
weights are replicated.  Exchange per optimiser step:
    all-reduce(sum) of the learner's flat buffer  [padded gradient | kl_sum | value_loss_sum | surrogate_sum | rows]
so that after it every rank holds the global gradient sum AND the global KL statistics; the LR schedule
(reference ppo.py:136-148) is then evaluated identically everywhere, gradients are scaled by 1/world inside the
Adam kernel, and parameters stay bit-identical across ranks without a broadcast.  Once per iteration a 3-double
all-reduce makes the advantage normalisation (rollout_storage.py:135-136) global.

Transport on GPUs: RCCL over xGMI INSIDE libhx.so (HxComm below; isaac_amd/csrc/hx_comm.hip, include/hx_ppo.h): the
learner enqueues the all-reduce on its own HIP stream between the backward kernels and the Adam kernel, the advantage
moments are all-reduced on the same stream, nothing synchronises with the host and torch.distributed is not involved
(its TCPStore carries the 128-byte RCCL unique id from rank 0 to the others, once).  TorchComm (torch.distributed,
backend "gloo") remains for the world_size-2 CPU tests of the orchestration (tests/test_parallel_cpu.py) and as
"gloo-staged" rehearsal of the real kernels with several ranks on one GPU.  xGMI sizing: 6.07 MB per step, ring
all-reduce moves 2*(7/8)*6.07 MB = 10.6 MB per link direction ~ 70 us at ~153 GB/s/link, < 1 % of a 61 440-row
minibatch step, so a single un-bucketed collective is the right granularity.
"""
import os

import numpy as np


class Comm:
    """world_size == 1 stand-in: every collective is the identity."""
    rank, world_size, local_rank = 0, 1, 0

    def broadcast_state(self, sd):
        return sd

    def alloc_grad_buffer(self, count):
        return None

    def all_reduce_grads(self, ptr, count, stream):
        pass

    def all_reduce_moments(self, ptr, stream):
        pass

    def barrier(self):
        pass

    def max_over_ranks(self, x):
        return x

    def sum_over_ranks(self, x):
        return x

    def sum_array_over_ranks(self, values):
        """Element-wise sum over ranks of a small float64 vector: ONE collective for all of an iteration's statistics."""
        return np.asarray(values, np.float64).copy()

    def close(self):
        pass


class TorchComm(Comm):
    def __init__(self, backend=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.world_size = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if backend is None:
            backend = os.environ.get("HX_DIST_BACKEND") or "gloo"
        if backend not in ("gloo", "gloo-staged"):
            raise ValueError("TorchComm carries gloo / gloo-staged only; the RCCL transport lives in libhx.so: "
                             "isaac_amd.parallel.HxComm (HX_DIST_BACKEND=rccl)")
        # "gloo-staged": gloo collectives on host copies of the library's DEVICE buffers.  A functional rehearsal of the
        # N > 1 path where RCCL cannot run (several ranks sharing one GPU); never the measured configuration.
        self.staged = (backend == "gloo-staged")
        if self.staged:
            backend = "gloo"
        self.backend = backend
        self.device_ptrs = self.staged      # do the library's pointers name device memory?
        self.device = torch.device("cpu")
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world_size)
        self._grad = None
        self._moments = None

    # ---- parameters start identical: rank 0's initialisation wins
    def broadcast_state(self, sd):
        out = {}
        for k, v in sd.items():
            t = self.torch.from_numpy(np.ascontiguousarray(v)).to(self.device)
            self.dist.broadcast(t, src=0)
            out[k] = t.cpu().numpy()
        return out

    def alloc_grad_buffer(self, count):
        """Flat fp32 buffer owned by torch so the collective can run on it in place (None: the library keeps its own
        device buffer and the collective is staged through the host)."""
        if self.staged:
            return None
        self._grad = self.torch.zeros(int(count), dtype=self.torch.float32, device=self.device)
        return self._grad.data_ptr()

    def all_reduce_grads(self, ptr, count, stream):
        if self.staged:
            from . import capi
            host = capi.download(ptr, np.float32, (int(count),), stream)
            t = self.torch.from_numpy(host)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
            capi.check(capi.lib().hx_memcpy_h2d(ptr, capi.ptr(np.ascontiguousarray(t.numpy())), int(count) * 4, stream), "h2d")
            return
        assert self._grad is not None and ptr == self._grad.data_ptr() and count == self._grad.numel()
        self.dist.all_reduce(self._grad, op=self.dist.ReduceOp.SUM)

    def all_reduce_moments(self, ptr, stream):
        """[sum, sum of squares, count] of the raw advantages (3 doubles) -> global moments on every rank."""
        import ctypes
        if self.device_ptrs:
            from . import capi
            m = capi.download(ptr, np.float64, (3,), stream)
        else:                                   # gloo: the pointer is host memory (CPU tests of the orchestration)
            m = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_double)), (3,)).copy()
        t = self.torch.from_numpy(m).to(self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        m = np.ascontiguousarray(t.cpu().numpy())
        if self.device_ptrs:
            from . import capi
            capi.check(capi.lib().hx_memcpy_h2d(ptr, capi.ptr(m), 24, stream), "h2d")
        else:
            ctypes.memmove(ptr, m.ctypes.data, 24)

    def barrier(self):
        self.dist.barrier()

    def max_over_ranks(self, x):
        t = self.torch.tensor([float(x)], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, x):
        t = self.torch.tensor([float(x)], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def sum_array_over_ranks(self, values):
        t = self.torch.tensor(np.asarray(values, np.float64), dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.numpy().copy()

    def close(self):
        """Tear the process group down."""
        if self.dist.is_initialized():
            self.dist.destroy_process_group()


class HxComm(Comm):
    """RCCL communicator owned by libhx.so.  in_library = True tells PPO that the library issues the collectives itself
    (hx_ppo_set_comm): PPO.update is then the same single C call as on one GPU."""
    in_library = True
    _created = 0

    def __init__(self, rank=None, world_size=None, local_rank=None):
        import ctypes as C
        from . import capi
        self.capi, self.C = capi, C
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else rank
        self.world_size = int(os.environ.get("WORLD_SIZE", "1")) if world_size is None else world_size
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0")) if local_rank is None else local_rank
        L = capi.lib()
        ndev = L.hx_device_count()
        if ndev < 1:
            raise RuntimeError("HxComm: no HIP device (the RCCL transport has no CPU form; HX_DIST_BACKEND=gloo for CPU rehearsals)")
        if self.local_rank >= ndev:
            # RCCL needs one GPU per rank: two ranks on one device make ncclCommInitRank fail or hang.  Only the gloo-staged
            # rehearsal transport may share a GPU.
            raise RuntimeError(f"HxComm: LOCAL_RANK {self.local_rank} but only {ndev} HIP device(s) visible: one rank per GPU "
                               "(HX_DIST_BACKEND=gloo-staged rehearses several ranks on one GPU)")
        capi.check(L.hx_set_device(self.local_rank), "hx_set_device")
        self._attached = []
        uid = (C.c_uint8 * 128)()

        def make_id():
            capi.check(L.hx_comm_get_unique_id(uid), "hx_comm_get_unique_id")
            return bytes(uid)
        raw = exchange_unique_id(self.rank, self.world_size, make_id, "hx_rccl_unique_id_%d" % HxComm._created)
        HxComm._created += 1           # one key per communicator of the job
        C.memmove(uid, raw, 128)
        h = C.c_void_p()
        capi.check(L.hx_comm_init(uid, self.rank, self.world_size, C.byref(h)), "hx_comm_init")
        self._h = h
        self._scratch = capi.DeviceBuffer(1024)

    def attach(self, alg):
        """Bind the learner: gradients / moments are all-reduced inside the library from now on; rank 0's parameters win."""
        import weakref
        self.capi.check(self.capi.lib().hx_ppo_set_comm(alg._h, self._h), "hx_ppo_set_comm")
        self.capi.check(self.capi.lib().hx_ppo_broadcast_params(alg._h, 0), "hx_ppo_broadcast_params")
        self._attached.append(weakref.ref(alg))

    def _reduce(self, values, op):
        a = np.ascontiguousarray(values, np.float64)
        assert a.nbytes <= self._scratch.nbytes
        L = self.capi.lib()
        self._scratch.upload(a)
        self.capi.check(L.hx_comm_all_reduce(self._h, self._scratch.ptr, a.size, 1, op, None), "hx_comm_all_reduce")
        self.capi.check(L.hx_comm_wait(self._h, None, 0.0), "hx_comm_wait")      # deadline, then abort: never an unbounded wait
        return self._scratch.download(np.float64, a.shape)

    def _reduce_scalar(self, x, op):
        return float(self._reduce([float(x)], op)[0])

    def sum_array_over_ranks(self, values):
        return self._reduce(values, 0)

    def barrier(self):
        self._reduce_scalar(0.0, 0)

    def max_over_ranks(self, x):
        return self._reduce_scalar(x, 1)

    def sum_over_ranks(self, x):
        return self._reduce_scalar(x, 0)

    def close(self):
        """Detach every learner that was given this communicator (they hold the raw pointer), then destroy it."""
        if getattr(self, "_h", None):
            for ref in self._attached:
                alg = ref()
                if alg is not None and getattr(alg, "_h", None):
                    self.capi.lib().hx_ppo_set_comm(alg._h, None)
            self._attached = []
            self.capi.lib().hx_comm_destroy(self._h)
            self._h = None


_stores = []


def exchange_unique_id(rank, world_size, make_id, key):
    """Host-side rendezvous of the RCCL unique id (plumbing only): rank 0 calls make_id() and publishes the bytes under `key`
    in a TCPStore on MASTER_ADDR:MASTER_PORT, the other ranks read them.  Under torch.distributed.run the launcher's agent
    already serves a store on MASTER_PORT (TORCHELASTIC_USE_AGENT_STORE=True): every rank is a client then; otherwise
    (bench.py's own launcher, hand-started ranks) rank 0 serves it.  world_size 1: no store."""
    if world_size <= 1:
        return make_id()
    from datetime import timedelta
    from torch.distributed import TCPStore
    agent_store = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "") == "True"
    addr, port = os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29511"))
    store = next((st for a, p_, st in _stores if (a, p_) == (addr, port)), None)
    if store is None:                  # one store per job: a second communicator reuses it (rank 0 cannot listen on the port twice)
        store = TCPStore(addr, port, world_size, rank == 0 and not agent_store, timeout=timedelta(seconds=300))
        _stores.append((addr, port, store))      # and the server side must outlive the clients' reads
    if rank == 0:
        raw = make_id()
        store.set(key, raw)
        return raw
    return bytes(store.get(key))


def init_comm(backend=None):
    """Comm for this process: under torch.distributed.run (WORLD_SIZE > 1) the in-library RCCL communicator on GPUs
    (HX_DIST_BACKEND=rccl, the default there), TorchComm for "gloo" / "gloo-staged"; the identity otherwise."""
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        backend = backend or os.environ.get("HX_DIST_BACKEND")
        if backend is None:
            from . import capi
            backend = "rccl" if capi.lib().hx_device_count() > 0 else "gloo"
        return HxComm() if backend in ("rccl", "nccl") else TorchComm(backend)
    return Comm()


def shard_envs(total_envs, rank, world_size):
    """Contiguous env shard [lo, hi) of rank `rank` (weak scaling keeps per-rank count fixed instead)."""
    per = total_envs // world_size
    lo = rank * per
    return lo, (lo + per if rank < world_size - 1 else total_envs)
