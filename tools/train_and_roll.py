#!/usr/bin/env python3
"""GPU: train a hector policy IN THIS simulator (ground plane, default HectorCfg otherwise), then evaluate it exactly like the
actors the reference ships (tools/actor_rollout.py protocol: 4096 robots, fixed command, 10 s).  Answers whether the gap
between commanded and walked speed of the PhysX-trained actor is a property of the physics or of the reward design: a policy
trained to convergence here can be held against the same yardstick.
usage: python tools/train_and_roll.py [iterations=1500] [out.jsonl]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from isaac_amd.envs import *  # noqa
from isaac_amd.utils import get_args, task_registry
from isaac_amd.utils.actor_eval import roll_actor

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
out = sys.argv[2] if len(sys.argv) > 2 else None
args = get_args(["--task=hector", "--headless", "--max_iterations", str(iters)])
env_cfg, train_cfg = task_registry.get_cfgs("hector")
env_cfg.terrain.mesh_type = "plane"
env, _ = task_registry.make_env(name="hector", args=args, env_cfg=env_cfg)
runner, train_cfg = task_registry.make_alg_runner(env=env, name=None, args=args, train_cfg=train_cfg, log_root=None)
runner.learn(num_learning_iterations=iters, init_at_random_ep_len=True)
sd = {k: v for k, v in runner.alg.actor_critic.state_dict().items() if k.startswith("actor.")}
info, _ = env.episode_stats()
runner.alg.close(); env.close()
rows = []
for vx in (0.1, 0.2, 0.3, 0.4, 0.5, 0.6):
    for push in (False, True):
        def edit(cfg, push=push):
            cfg.domain_rand.push_robots = push
        r = roll_actor(sd, num_envs=4096, steps=1000, command=(vx, 0.0, 0.0, 0.0), cfg_edit=edit)
        r.update(actor=f"trained here, {iters} iterations on the plane", pushes=push)
        rows.append(r)
        print(json.dumps(r), flush=True)
if out:
    with open(out, "w") as f:
        for r in rows:
            f.write(json.dumps(r) + "\n")
