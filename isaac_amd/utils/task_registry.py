"""name -> (env class, env cfg, train cfg) registry with the reference's API
(humanoid/utils/task_registry.py:44-163): register / get_task_class / get_cfgs / make_env / make_alg_runner."""
import os
from datetime import datetime

from .. import LEGGED_GYM_ROOT_DIR
from ..algo.on_policy_runner import OnPolicyRunner
from .helpers import class_to_dict, get_args, get_load_path, parse_sim_params, set_seed, update_cfg_from_args

_RUNNERS = {"OnPolicyRunner": OnPolicyRunner}


class TaskRegistry:
    def __init__(self):
        self.task_classes, self.env_cfgs, self.train_cfgs = {}, {}, {}

    def register(self, name, task_class, env_cfg, train_cfg):
        self.task_classes[name] = task_class
        self.env_cfgs[name] = env_cfg
        self.train_cfgs[name] = train_cfg

    def get_task_class(self, name):
        return self.task_classes[name]

    def get_cfgs(self, name):
        train_cfg, env_cfg = self.train_cfgs[name], self.env_cfgs[name]
        env_cfg.seed = train_cfg.seed          # task_registry.py:62
        return env_cfg, train_cfg

    def make_env(self, name, args=None, env_cfg=None, comm=None):
        if args is None:
            args = get_args()
        if name not in self.task_classes:
            raise ValueError(f"Task with name: {name} was not registered")
        task_class = self.get_task_class(name)
        if env_cfg is None:
            env_cfg, _ = self.get_cfgs(name)
        env_cfg, _ = update_cfg_from_args(env_cfg, None, args)
        # data parallel: every rank owns its own shard of envs with its own random stream
        rank = 0 if comm is None else comm.rank
        env_cfg.seed = set_seed(env_cfg.seed + rank if env_cfg.seed != -1 else -1)
        sim_params = parse_sim_params(args, {"sim": class_to_dict(env_cfg.sim)})
        env = task_class(cfg=env_cfg, sim_params=sim_params, physics_engine=args.physics_engine,
                         sim_device=args.sim_device, headless=args.headless)
        self.env_cfg_for_wandb = env_cfg
        return env, env_cfg

    def make_alg_runner(self, env, name=None, args=None, train_cfg=None, log_root="default", comm=None):
        if args is None:
            args = get_args()
        if train_cfg is None:
            if name is None:
                raise ValueError("Either 'name' or 'train_cfg' must be not None")
            _, train_cfg = self.get_cfgs(name)
        elif name is not None:
            print(f"'train_cfg' provided -> Ignoring 'name={name}'")
        _, train_cfg = update_cfg_from_args(None, train_cfg, args)
        if log_root == "default":
            log_root = os.path.join(LEGGED_GYM_ROOT_DIR, "logs", train_cfg.runner.experiment_name)
            log_dir = os.path.join(log_root, datetime.now().strftime("%b%d_%H-%M-%S") + "_" + train_cfg.runner.run_name)
        elif log_root is None:
            log_dir = None
        else:
            log_dir = os.path.join(log_root, datetime.now().strftime("%b%d_%H-%M-%S") + "_" + train_cfg.runner.run_name)
        train_cfg_dict = class_to_dict(train_cfg)
        env_cfg_dict = class_to_dict(getattr(self, "env_cfg_for_wandb", None)) if hasattr(self, "env_cfg_for_wandb") else {}
        all_cfg = {**train_cfg_dict, **(env_cfg_dict or {})}
        runner_class = _RUNNERS[train_cfg_dict["runner_class_name"]]
        runner = runner_class(env, all_cfg, log_dir, device=args.rl_device, comm=comm)
        if train_cfg.runner.resume:
            resume_path = get_load_path(log_root, load_run=train_cfg.runner.load_run, checkpoint=train_cfg.runner.checkpoint)
            print(f"Loading model from: {resume_path}")
            runner.load(resume_path, load_optimizer=False)
        return runner, train_cfg


task_registry = TaskRegistry()
