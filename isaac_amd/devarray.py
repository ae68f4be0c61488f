"""DeviceArray: a typed view of device memory owned by libhx (no tensor library required).

Exposes `__cuda_array_interface__` (v2), so `torch.as_tensor(x, device='cuda')` or cupy can alias it
without a copy; `.numpy()` downloads.  This is what the VecEnv attributes (obs_buf, rew_buf, ...) are.
"""
import numpy as np

from . import capi


class DeviceArray:
    def __init__(self, ptr, shape, dtype=np.float32, strides=None, stream=None, owner=None):
        self.ptr = int(ptr)
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.strides = None if strides is None else tuple(int(s) for s in strides)   # bytes
        self.stream = stream
        self._owner = owner

    @property
    def __cuda_array_interface__(self):
        return {"shape": self.shape, "typestr": self.dtype.str, "data": (self.ptr, False), "version": 2,
                "strides": self.strides}

    def data_ptr(self):
        return self.ptr

    def numpy(self):
        """Synchronising download (for tests, logging and checkpoints)."""
        if self.strides is None:
            return capi.download(self.ptr, self.dtype, self.shape, self.stream)
        # row-padded 2-D buffers: download the padded rows, then slice
        assert len(self.shape) == 2 and self.strides[1] == self.dtype.itemsize
        ld = self.strides[0] // self.dtype.itemsize
        full = capi.download(self.ptr, self.dtype, (self.shape[0], ld), self.stream)
        return np.ascontiguousarray(full[:, :self.shape[1]])

    def cpu(self):
        return self.numpy()

    def __repr__(self):
        return f"DeviceArray(ptr=0x{self.ptr:x}, shape={self.shape}, dtype={self.dtype})"


def device_pointer(x, staging=None):
    """Pointer of a device-resident array-like (DeviceArray / torch.cuda tensor / cupy), or upload a host array."""
    if isinstance(x, DeviceArray):
        return x.ptr, None
    cai = getattr(x, "__cuda_array_interface__", None)
    if cai is not None:
        return int(cai["data"][0]), None
    if hasattr(x, "detach") and hasattr(x, "cpu"):       # a CPU torch tensor
        x = x.detach().cpu().numpy()
    arr = np.ascontiguousarray(x)
    buf = capi.DeviceBuffer.from_host(arr)
    return buf.ptr, buf
