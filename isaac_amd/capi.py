"""ctypes binding of libhx.so (include/hx_sim.h, include/hx_ppo.h).

This is the "thin C-ABI/ctypes layer" of the north star: no arithmetic lives here.  If the shared
library is missing or fails to load the import raises -- there is no CPU or PyTorch fallback.
"""
import ctypes as C
import os

# Kernel arguments in device memory instead of host-coherent memory: the rollout is a chain of ~180 dependent launches per
# iteration and every launch reads its arguments first; with host-side kernargs the chain is 0.6 ms per iteration slower
# (profiles/r02_e_critic_chunk.txt).  The default of recent ROCm releases on this part; made explicit here, before the HIP
# runtime of this process initialises.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhx.so")

NUM_DOF = 10
MAX_DOF, MAX_OBS_FRAME = 18, 65          # hector_full (include/hx_sim.h)
OBS_FRAME, PRIV_FRAME, FRAME_STACK = 41, 70, 15
NUM_OBS, NUM_PRIV = 615, 1050
OBS_LD, PRIV_LD = 616, 1052
NUM_REWARDS = 22
RP_SIZE = 75
RP = dict(delay=0, act_noise=1, cmd_a=11, push=14, reset_q=19, reset_xy=29, cmd_b=31, obs_noise=34, level=75)
REWARD_NAMES = ["action_smoothness", "base_acc", "base_height", "collision", "default_joint_pos", "dof_acc",
                "dof_vel", "feet_air_time", "feet_clearance", "feet_contact_forces", "feet_contact_number",
                "feet_distance", "foot_slip", "joint_pos", "knee_distance", "low_speed", "orientation", "torques",
                "track_vel_hard", "tracking_ang_vel", "tracking_lin_vel", "vel_mismatch_exp"]
(BUF_OBS, BUF_PRIV, BUF_REW, BUF_RESET, BUF_TIMEOUT, BUF_TIMEOUT_VISIBLE, BUF_EP_LEN, BUF_COMMANDS, BUF_TORQUES,
 BUF_CONTACT, BUF_BODY_STATE, BUF_EPISODE_SUMS, BUF_FEET_AIR_TIME, BUF_FEET_HEIGHT, BUF_NUM_RESET) = range(15)
# enum hx_ppo_buffer_id (include/hx_ppo.h)
(PPO_BUF_ACTIONS, PPO_BUF_VALUES, PPO_BUF_LOGP, PPO_BUF_MU, PPO_BUF_REWARDS, PPO_BUF_RETURNS, PPO_BUF_ADVANTAGES, PPO_BUF_GRADS,
 PPO_BUF_PERM, PPO_BUF_OBS, PPO_BUF_PRIV, PPO_BUF_DONES, PPO_BUF_TIMEOUTS) = range(13)


class SimCfg(C.Structure):
    _fields_ = [
        ("num_envs", C.c_int32), ("decimation", C.c_int32), ("sim_dt", C.c_float), ("gravity_z", C.c_float),
        ("action_scale", C.c_float), ("clip_actions", C.c_float), ("clip_observations", C.c_float),
        ("default_dof_pos", C.c_float * MAX_DOF), ("p_gains", C.c_float * MAX_DOF), ("d_gains", C.c_float * MAX_DOF),
        ("torque_limits", C.c_float * MAX_DOF),
        ("action_delay", C.c_float), ("action_noise", C.c_float), ("add_noise", C.c_int32), ("noise_level", C.c_float),
        ("noise_scale_vec", C.c_float * MAX_OBS_FRAME),
        ("push_robots", C.c_int32), ("push_interval", C.c_int32), ("max_push_vel_xy", C.c_float),
        ("max_push_ang_vel", C.c_float),
        ("resample_interval", C.c_int32), ("heading_command", C.c_int32), ("cmd_range", (C.c_float * 2) * 4),
        ("obs_scale_lin_vel", C.c_float), ("obs_scale_ang_vel", C.c_float), ("obs_scale_dof_pos", C.c_float),
        ("obs_scale_dof_vel", C.c_float), ("obs_scale_quat", C.c_float),
        ("max_episode_length", C.c_float), ("max_episode_length_s", C.c_float), ("env_dt", C.c_float),
        ("base_init_state", C.c_float * 13), ("custom_origins", C.c_int32),
        ("reward_scale", C.c_float * NUM_REWARDS), ("only_positive_rewards", C.c_int32),
        ("base_height_target", C.c_float), ("min_dist", C.c_float), ("max_dist", C.c_float),
        ("target_joint_pos_scale", C.c_float), ("target_feet_height", C.c_float),
        ("cycle_time", C.c_float), ("tracking_sigma", C.c_float), ("max_contact_force", C.c_float),
        ("contact_kn", C.c_float), ("contact_dn", C.c_float), ("friction_veps", C.c_float),
        ("limit_k", C.c_float), ("limit_d", C.c_float), ("terrain_mu", C.c_float),
        ("env_id_offset", C.c_int32), ("num_dof", C.c_int32),
        ("max_depenetration_velocity", C.c_float), ("contact_offset", C.c_float), ("rest_offset", C.c_float),
        ("self_collisions", C.c_int32),
    ]


class PpoCfg(C.Structure):
    _fields_ = [
        ("num_envs", C.c_int32), ("num_steps", C.c_int32), ("num_obs", C.c_int32), ("num_priv", C.c_int32),
        ("num_actions", C.c_int32), ("actor_hidden", C.c_int32 * 3), ("critic_hidden", C.c_int32 * 3),
        ("num_learning_epochs", C.c_int32), ("num_mini_batches", C.c_int32),
        ("clip_param", C.c_float), ("gamma", C.c_float), ("lam", C.c_float), ("value_loss_coef", C.c_float),
        ("entropy_coef", C.c_float), ("learning_rate", C.c_float), ("max_grad_norm", C.c_float),
        ("use_clipped_value_loss", C.c_int32), ("adaptive_schedule", C.c_int32), ("desired_kl", C.c_float),
        ("init_noise_std", C.c_float), ("obs_ld", C.c_int32), ("priv_ld", C.c_int32),
        ("obs_frame", C.c_int32), ("priv_frame", C.c_int32), ("obs_stack", C.c_int32), ("priv_stack", C.c_int32),
    ]


class ProfRow(C.Structure):          # include/hx_lab.h hx_prof_row
    _fields_ = [("symbol", C.c_char * 128), ("ms", C.c_double), ("launches", C.c_int64), ("flops", C.c_double)]


_lib = None


def lib():
    """Load libhx.so (built by __graft_entry__.build() / isaac_amd.build).  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(the HIP extension is mandatory; there is no fallback path)")
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, f32p, i32p = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32)
    L.hx_last_error.restype = C.c_char_p
    L.hx_version.restype = C.c_int
    L.hx_build_id.restype = C.c_char_p
    L.hx_sync.argtypes = [vp]
    L.hx_sim_create.argtypes = [C.POINTER(SimCfg), vp, vp, vp, vp, C.c_uint64, vp, C.POINTER(vp)]
    L.hx_sim_destroy.argtypes = [vp]
    L.hx_sim_destroy.restype = None
    L.hx_sim_reset_all.argtypes = [vp, vp]
    L.hx_sim_step.argtypes = [vp, vp, vp]
    L.hx_sim_step_ex.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    L.hx_sim_set_pause_word.argtypes = [vp, vp]
    L.hx_sim_buffer.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.hx_sim_get_state.argtypes = [vp, vp, vp, vp]
    L.hx_sim_set_state.argtypes = [vp, vp, vp, vp]
    L.hx_sim_set_episode_length.argtypes = [vp, vp]
    L.hx_sim_set_step_counter.argtypes = [vp, C.c_int64]
    L.hx_sim_set_commands.argtypes = [vp, vp]
    L.hx_sim_get_base_velocities.argtypes = [vp, vp, vp]
    L.hx_sim_set_terrain.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float]
    L.hx_sim_set_terrain_options.argtypes = [vp, C.c_int32]
    L.hx_sim_set_terrain_curriculum.argtypes = [vp, vp, C.c_int32, C.c_int32, vp, vp, C.c_float, C.c_float]
    L.hx_sim_get_terrain_levels.argtypes = [vp, vp]
    L.hx_sim_episode_stats.argtypes = [vp, vp, vp]
    L.hx_sim_stream.argtypes = [vp]
    L.hx_sim_prof.argtypes = [vp, C.c_int, vp]
    L.hx_sim_time.argtypes = [vp, C.c_int, vp]
    L.hx_sim_stream.restype = vp
    # generic device memory helpers
    L.hx_malloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    L.hx_free.argtypes = [vp]
    L.hx_memcpy_h2d.argtypes = [vp, vp, C.c_size_t, vp]
    L.hx_memcpy_d2h.argtypes = [vp, vp, C.c_size_t, vp]
    L.hx_memcpy_d2d.argtypes = [vp, vp, C.c_size_t, vp]
    L.hx_device_count.restype = C.c_int
    L.hx_set_device.argtypes = [C.c_int]
    # learner
    L.hx_ppo_create.argtypes = [C.POINTER(PpoCfg), vp, vp, C.POINTER(vp)]
    L.hx_ppo_destroy.argtypes = [vp]
    L.hx_ppo_destroy.restype = None
    L.hx_ppo_stream.argtypes = [vp]
    L.hx_ppo_stream.restype = vp
    L.hx_ppo_num_params.argtypes = [vp]
    L.hx_ppo_num_params.restype = C.c_int64
    L.hx_ppo_set_params_h.argtypes = [vp, vp]
    L.hx_ppo_get_params_h.argtypes = [vp, vp]
    L.hx_ppo_set_compute_dtype.argtypes = [vp, C.c_int]
    L.hx_ppo_set_opt_state_h.argtypes = [vp, vp, vp, C.c_int64]
    L.hx_ppo_get_opt_state_h.argtypes = [vp, vp, vp, C.POINTER(C.c_int64)]
    L.hx_ppo_set_seed.argtypes = [vp, C.c_uint64, C.c_uint64]
    L.hx_ppo_get_rng_state.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.hx_ppo_set_rng_state.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.hx_ppo_act.argtypes = [vp, vp, vp, vp, C.POINTER(vp)]
    L.hx_ppo_process_step.argtypes = [vp, vp, vp, vp]
    L.hx_ppo_compute_returns.argtypes = [vp, vp]
    L.hx_ppo_act_range.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, vp, C.POINTER(vp)]
    L.hx_ppo_process_step_range.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, vp, C.c_int]
    L.hx_ppo_last_values_range.argtypes = [vp, vp, C.c_int, C.c_int, vp]
    L.hx_ppo_adv_moments.argtypes = [vp, C.POINTER(vp)]
    L.hx_ppo_adv_normalize.argtypes = [vp]
    L.hx_comm_get_unique_id.argtypes = [vp]
    L.hx_comm_init.argtypes = [vp, C.c_int, C.c_int, C.POINTER(vp)]
    L.hx_comm_destroy.argtypes = [vp]
    L.hx_comm_destroy.restype = None
    L.hx_comm_rank.argtypes = [vp]
    L.hx_comm_world.argtypes = [vp]
    L.hx_comm_all_reduce.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_int, vp]
    L.hx_comm_broadcast.argtypes = [vp, vp, C.c_size_t, C.c_int, vp]
    L.hx_comm_wait.argtypes = [vp, vp, C.c_double]
    L.hx_comm_library_info.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.hx_ppo_set_comm.argtypes = [vp, vp]
    L.hx_ppo_set_row_base.argtypes = [vp, C.c_uint32]
    L.hx_ppo_broadcast_params.argtypes = [vp, C.c_int]
    L.hx_ppo_update_begin.argtypes = [vp, vp]
    L.hx_ppo_minibatch_backward.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_int64)]
    L.hx_ppo_minibatch_step.argtypes = [vp, C.c_float]
    L.hx_ppo_update_end.argtypes = [vp, vp]
    L.hx_ppo_update.argtypes = [vp, vp, vp]
    L.hx_rollout.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int, C.c_int]
    L.hx_ppo_storage_rows.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp]
    L.hx_ppo_buffer.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.hx_ppo_get_lr.argtypes = [vp, vp]
    L.hx_ppo_set_lr.argtypes = [vp, C.c_float]
    L.hx_ppo_inference.argtypes = [vp, vp, C.c_int, vp]
    L.hx_ppo_prof_begin.argtypes = [vp, C.c_char_p, C.c_int]
    L.hx_ppo_prof_end.argtypes = [vp, vp, C.c_int, C.POINTER(C.c_int)]
    L.hx_wgrad_plan_describe.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_longlong), C.c_int,
                                         C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_longlong)]
    L.hx_ppo_actor_stamps.argtypes = [vp, vp, C.c_int]
    L.hx_ppo_pause_words.argtypes = [vp, vp]
    L.hx_sim_prof_waves.argtypes = [vp, vp, C.c_int]
    L.hx_sim_prof_last.argtypes = [vp, vp, C.c_int]
    L.hx_ppo_gemm_bench.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    L.hx_ppo_gemm_test.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, C.c_int, vp, vp, C.c_int, vp, vp]
    L.hx_ppo_wgrad_multi_test.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp),
                                          C.c_int, C.POINTER(C.c_int), vp]
    _lib = L
    return L


def check(rc, what=""):
    if rc != 0:
        raise RuntimeError(f"{what} failed rc={rc}: {lib().hx_last_error().decode()}")


def farr(x):
    return np.ascontiguousarray(x, dtype=np.float32)


def ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class DeviceBuffer:
    """Device memory owned by the library's allocator; plain bytes, no torch involved."""

    def __init__(self, nbytes, stream=None):
        self.nbytes = int(nbytes)
        self.stream = stream
        p = C.c_void_p()
        check(lib().hx_malloc(self.nbytes, C.byref(p)), "hx_malloc")
        self.ptr = p.value

    @classmethod
    def from_host(cls, arr, stream=None):
        arr = np.ascontiguousarray(arr)
        b = cls(arr.nbytes, stream)
        b.upload(arr)
        return b

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        check(lib().hx_memcpy_h2d(self.ptr, ptr(arr), arr.nbytes, self.stream), "h2d")

    def download(self, dtype, shape):
        out = np.empty(shape, dtype)
        assert out.nbytes <= self.nbytes
        check(lib().hx_memcpy_d2h(ptr(out), self.ptr, out.nbytes, self.stream), "d2h")
        return out

    def free(self):
        if self.ptr:
            lib().hx_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def download(dptr, dtype, shape, stream=None):
    out = np.empty(shape, dtype)
    check(lib().hx_memcpy_d2h(ptr(out), dptr, out.nbytes, stream), "d2h")
    return out
