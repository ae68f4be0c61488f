#!/usr/bin/env python3
"""Training run with config overrides for experiments (GPU box only).
usage: train_variant.py <iterations> <run_name> [task=hector_full] [key=value ...]   e.g. terrain.mesh_type=plane domain_rand.push_robots=False"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isaac_amd.envs import *  # noqa
from isaac_amd.utils import get_args, task_registry

iters, run = int(sys.argv[1]), sys.argv[2]
task = "hector"
if len(sys.argv) > 3 and sys.argv[3].startswith("task="):
    task = sys.argv.pop(3)[len("task="):]
args = get_args([f"--task={task}", "--headless", "--run_name", run, "--max_iterations", str(iters)])
env_cfg, train_cfg = task_registry.get_cfgs(task)
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    obj = env_cfg
    if k.startswith("train."):            # e.g. train.algorithm.mlp_dtype=bf16
        obj, k = train_cfg, k[len("train."):]
    parts = k.split(".")
    for p in parts[:-1]:
        obj = getattr(obj, p)
    old = getattr(obj, parts[-1], "")
    setattr(obj, parts[-1], type(old)(eval(v)) if not isinstance(old, str) else v)
    print("override", k, "=", getattr(obj, parts[-1]))
env, _ = task_registry.make_env(name=task, args=args, env_cfg=env_cfg)
runner, train_cfg = task_registry.make_alg_runner(env=env, name=None, args=args, train_cfg=train_cfg)
runner.learn(num_learning_iterations=iters, init_at_random_ep_len=True)
