"""Data parallelism for the hector hot path: one process per GPU, envs sharded, ONE collective per optimiser step.

The reference is single-process (an unused --horovod flag is its only trace of DP).  The path
shards naturally -- environments never interact -- so rank r owns its own simulator + rollout storage and the
weights are replicated.  Exchange per optimiser step:
    all-reduce(sum) of the learner's flat buffer  [padded gradient | kl_sum | value_loss_sum | surrogate_sum | rows]
so that after it every rank holds the global gradient sum AND the global KL statistics; the LR schedule
(reference ppo.py:136-148) is then evaluated identically everywhere, gradients are scaled by 1/world inside the
Adam kernel, and parameters stay bit-identical across ranks without a broadcast.  Once per iteration a 3-double
all-reduce makes the advantage normalisation (rollout_storage.py:135-136) global.

Transport: torch.distributed -- backend "nccl" (= RCCL over xGMI on ROCm) on GPUs; backend "gloo" for the
world_size-2 CPU tests of this orchestration (tests/test_parallel_cpu.py).  The gradient buffer is allocated
by torch when world_size > 1 and handed to the library as an external buffer (hx_ppo_create ext_grad_buffer),
so the collective runs on the tensor in place with no staging copy.  xGMI sizing: 6.07 MB per step, ring
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
            backend = os.environ.get("HX_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        # "gloo-staged": gloo collectives on host copies of the library's DEVICE buffers.  A functional rehearsal of the
        # N > 1 path where RCCL cannot run (several ranks sharing one GPU); never the measured configuration.
        self.staged = (backend == "gloo-staged")
        if self.staged:
            backend = "gloo"
        self.backend = backend
        self.device_ptrs = (backend == "nccl") or self.staged      # do the library's pointers name device memory?
        if backend == "nccl":
            torch.cuda.set_device(self.local_rank)
            self.device = torch.device("cuda", self.local_rank)
        else:
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
        if self.backend == "nccl":
            self.torch.cuda.current_stream().synchronize()      # the fill ran on torch's stream; the library uses its own
        return self._grad.data_ptr()

    def _sync_streams(self, stream):
        """The library launches on its own HIP stream; torch's collectives are ordered against torch's current
        stream.  Make each wait for the other with a full stream sync (two per optimiser step, ~10 us each,
        against a multi-millisecond minibatch)."""
        if self.backend == "nccl":
            from . import capi
            capi.check(capi.lib().hx_sync(stream), "hx_sync")

    def all_reduce_grads(self, ptr, count, stream):
        if self.staged:
            from . import capi
            host = capi.download(ptr, np.float32, (int(count),), stream)
            t = self.torch.from_numpy(host)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
            capi.check(capi.lib().hx_memcpy_h2d(ptr, capi.ptr(np.ascontiguousarray(t.numpy())), int(count) * 4, stream), "h2d")
            return
        assert self._grad is not None and ptr == self._grad.data_ptr() and count == self._grad.numel()
        self._sync_streams(stream)
        self.dist.all_reduce(self._grad, op=self.dist.ReduceOp.SUM)
        if self.backend == "nccl":
            self.torch.cuda.current_stream().synchronize()

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

    def close(self):
        """Tear the process group down (the nccl backend warns at exit otherwise)."""
        if self.dist.is_initialized():
            self.dist.destroy_process_group()


def init_comm(backend=None):
    """Comm for this process: TorchComm under torch.distributed.run (WORLD_SIZE > 1), identity otherwise."""
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        return TorchComm(backend)
    return Comm()


def shard_envs(total_envs, rank, world_size):
    """Contiguous env shard [lo, hi) of rank `rank` (weak scaling keeps per-rank count fixed instead)."""
    per = total_envs // world_size
    lo = rank * per
    return lo, (lo + per if rank < world_size - 1 else total_envs)
