"""CLI / config helpers with the reference's names (humanoid/utils/helpers.py).

`get_args` re-provides the flags the reference gets from `isaacgym.gymutil.parse_arguments`
(--sim_device, --pipeline, --graphics_device_id, --physx/--flex, --num_threads, --subscenes, --slices)
next to its own (--task, --resume, --experiment_name, --run_name, --load_run, --checkpoint, --headless,
--horovod, --rl_device, --num_envs, --seed, --max_iterations; helpers.py:161-239).  Flags that only make
sense for PhysX are accepted and ignored.
"""
import argparse
import os
import random

import numpy as np

from ..cfgtools import class_to_dict  # noqa: F401  (re-exported, helpers.py:43)


class SimParams:
    """Stand-in for gymapi.SimParams: the fields the env reads (dt, substeps, gravity, use_gpu_pipeline)."""

    def __init__(self):
        self.dt = 1.0 / 60.0
        self.substeps = 1
        self.use_gpu_pipeline = True
        self.gravity = [0.0, 0.0, -9.81]
        self.physx = type("physx", (), {})()


def update_class_from_dict(obj, d):
    for key, val in d.items():
        attr = getattr(obj, key, None)
        if isinstance(attr, type) or (hasattr(attr, "__dict__") and isinstance(val, dict)):
            update_class_from_dict(attr, val)
        else:
            setattr(obj, key, val)


def set_seed(seed):
    if seed == -1:
        seed = np.random.randint(0, 10000)
    print("Setting seed: {}".format(seed))
    random.seed(seed)
    np.random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    try:
        import torch
        torch.manual_seed(seed)
    except ImportError:                                  # pragma: no cover
        pass
    return seed


def parse_sim_params(args, cfg):
    sp = SimParams()
    sp.use_gpu_pipeline = getattr(args, "use_gpu_pipeline", True)
    for k, v in cfg.get("sim", {}).items():
        if k == "physx":
            for kk, vv in v.items():
                setattr(sp.physx, kk, vv)
        else:
            setattr(sp, k, v)
    if getattr(args, "num_threads", 0) and args.num_threads > 0:
        sp.physx.num_threads = args.num_threads
    return sp


def get_load_path(root, load_run=-1, checkpoint=-1):
    try:
        runs = sorted(os.listdir(root))
        if "exported" in runs:
            runs.remove("exported")
        last_run = os.path.join(root, runs[-1])
    except Exception:
        raise ValueError("No runs in this directory: " + root)
    load_run = last_run if load_run in (-1, "-1") else os.path.join(root, load_run)
    if checkpoint in (-1, "-1"):
        models = [f for f in os.listdir(load_run) if "model" in f]
        models.sort(key=lambda m: "{0:0>15}".format(m))
        model = models[-1]
    else:
        model = "model_{}.pt".format(checkpoint)
    return os.path.join(load_run, model)


def update_cfg_from_args(env_cfg, cfg_train, args):
    if env_cfg is not None and args.num_envs is not None:
        env_cfg.env.num_envs = args.num_envs
    if cfg_train is not None:
        if args.seed is not None:
            cfg_train.seed = args.seed
        if args.max_iterations is not None:
            cfg_train.runner.max_iterations = args.max_iterations
        if args.resume:
            cfg_train.runner.resume = args.resume
        for name in ("experiment_name", "run_name", "load_run", "checkpoint"):
            if getattr(args, name) is not None:
                setattr(cfg_train.runner, name, getattr(args, name))
    return env_cfg, cfg_train


def get_args(argv=None):
    p = argparse.ArgumentParser(description="RL Policy")
    p.add_argument("--task", type=str, default="hector")
    p.add_argument("--resume", action="store_true", default=False)
    p.add_argument("--experiment_name", type=str)
    p.add_argument("--run_name", type=str)
    p.add_argument("--load_run", type=str)
    p.add_argument("--checkpoint", type=int)
    p.add_argument("--headless", action="store_true", default=False)
    p.add_argument("--horovod", action="store_true", default=False)
    p.add_argument("--rl_device", type=str, default="cuda:0")
    p.add_argument("--num_envs", type=int)
    p.add_argument("--seed", type=int)
    p.add_argument("--max_iterations", type=int)
    # play script only (no reference counterpart: the reference edits constants in play.py)
    p.add_argument("--onnx", type=str, default=None, help="play: load the actor from this ONNX file instead of a checkpoint")
    p.add_argument("--play_steps", type=int, default=1200, help="play: env steps to roll (reference stop_state_log)")
    p.add_argument("--play_out", type=str, default=None, help="play: directory for traces and exported policies")
    # flags isaacgym.gymutil.parse_arguments provides
    p.add_argument("--sim_device", type=str, default="cuda:0")
    p.add_argument("--pipeline", type=str, default="gpu")
    p.add_argument("--graphics_device_id", type=int, default=0)
    p.add_argument("--physx", action="store_true", default=True)
    p.add_argument("--flex", action="store_true", default=False)
    p.add_argument("--num_threads", type=int, default=0)
    p.add_argument("--subscenes", type=int, default=0)
    p.add_argument("--slices", type=int, default=None)
    args = p.parse_args(argv)
    dev = args.sim_device.split(":")
    args.sim_device_type = dev[0]
    args.compute_device_id = int(dev[1]) if len(dev) > 1 else 0
    args.sim_device_id = args.compute_device_id
    args.use_gpu = args.sim_device_type == "cuda"
    args.use_gpu_pipeline = args.pipeline in ("gpu", "cuda")
    args.physics_engine = 1      # SIM_PHYSX placeholder: there is one simulator here
    # one process per GPU under torch.distributed.run: the local rank selects the device
    lr = os.environ.get("LOCAL_RANK")
    if lr is not None and args.sim_device_type == "cuda":
        args.compute_device_id = args.sim_device_id = int(lr)
        args.sim_device = f"cuda:{lr}"
        args.rl_device = f"cuda:{lr}"
    return args


def export_policy_as_onnx(actor_critic, path, name="locomotion_net.onnx"):
    """The `torch.onnx.export(actor, obs, "locomotion_net.onnx", opset_version=11, input_names=['obs'],
    output_names=['action'])` of reference play.py:89-98, written without the onnx package (utils/onnx_io.py)."""
    from .onnx_io import save_actor
    os.makedirs(path, exist_ok=True)
    sd = actor_critic.state_dict()
    layers = [(sd[f"actor.{2 * i}.weight"], sd[f"actor.{2 * i}.bias"]) for i in range(len(actor_critic.actor_hidden_dims) + 1)]
    return save_actor(os.path.join(path, name), layers)


def export_policy_as_jit(actor_critic, path):
    """reference helpers.py:242-247: TorchScript copy of the actor (CPU), `policy_1.pt`."""
    import torch
    os.makedirs(path, exist_ok=True)
    sd = actor_critic.state_dict()
    dims = [actor_critic.num_actor_obs, *actor_critic.actor_hidden_dims, actor_critic.num_actions]
    layers = []
    for i in range(4):
        layers.append(torch.nn.Linear(dims[i], dims[i + 1]))
        if i < 3:
            layers.append(torch.nn.ELU())
    model = torch.nn.Sequential(*layers)
    model.load_state_dict({k[len("actor."):]: torch.from_numpy(v) for k, v in sd.items() if k.startswith("actor.")})
    torch.jit.script(model).save(os.path.join(path, "policy_1.pt"))
    return os.path.join(path, "policy_1.pt")
