"""Restatement of the published semantics of isaacgym.torch_utils helpers that the reference uses
(`from isaacgym.torch_utils import *`, reference legged_robot.py:36).  xyzw quaternions."""
import numpy as np
import torch

_draw_log = None   # set by the fixture generator to record every random draw


def to_torch(x, dtype=torch.float, device="cpu", requires_grad=False):
    return torch.tensor(x, dtype=dtype, device=device, requires_grad=requires_grad)


def normalize(x, eps: float = 1e-9):
    return x / x.norm(p=2, dim=-1).clamp(min=eps, max=None).unsqueeze(-1)


def quat_apply(a, b):
    shape = b.shape
    a = a.reshape(-1, 4)
    b = b.reshape(-1, 3)
    xyz = a[:, :3]
    t = xyz.cross(b, dim=-1) * 2
    return (b + a[:, 3:] * t + xyz.cross(t, dim=-1)).view(shape)


def quat_rotate_inverse(q, v):
    shape = q.shape
    q_w = q[:, -1]
    q_vec = q[:, :3]
    a = v * (2.0 * q_w ** 2 - 1.0).unsqueeze(-1)
    b = torch.cross(q_vec, v, dim=-1) * q_w.unsqueeze(-1) * 2.0
    c = q_vec * torch.bmm(q_vec.view(shape[0], 1, 3), v.view(shape[0], 3, 1)).squeeze(-1) * 2.0
    return a - b + c


def get_euler_xyz(q):
    qx, qy, qz, qw = 0, 1, 2, 3
    sinr_cosp = 2.0 * (q[:, qw] * q[:, qx] + q[:, qy] * q[:, qz])
    cosr_cosp = q[:, qw] * q[:, qw] - q[:, qx] * q[:, qx] - q[:, qy] * q[:, qy] + q[:, qz] * q[:, qz]
    roll = torch.atan2(sinr_cosp, cosr_cosp)
    sinp = 2.0 * (q[:, qw] * q[:, qy] - q[:, qz] * q[:, qx])
    pitch = torch.where(torch.abs(sinp) >= 1, torch.sign(sinp) * (np.pi / 2.0), torch.asin(sinp))
    siny_cosp = 2.0 * (q[:, qw] * q[:, qz] + q[:, qx] * q[:, qy])
    cosy_cosp = q[:, qw] * q[:, qw] + q[:, qx] * q[:, qx] - q[:, qy] * q[:, qy] - q[:, qz] * q[:, qz]
    yaw = torch.atan2(siny_cosp, cosy_cosp)
    return roll % (2 * np.pi), pitch % (2 * np.pi), yaw % (2 * np.pi)


def torch_rand_float(lower, upper, shape, device):
    r = (upper - lower) * torch.rand(*shape, device=device) + lower
    if _draw_log is not None:
        _draw_log.append(("uniform", float(lower), float(upper), r.detach().clone()))
    return r


def get_axis_params(value, axis_idx, x_value=0., dtype=float, n_dims=3):
    zs = np.zeros((n_dims,))
    assert axis_idx < n_dims
    zs[axis_idx] = 1.
    params = np.where(zs == 1., value, zs)
    params[0] = x_value
    return list(params.astype(dtype))
