"""`python -m isaac_amd.scripts.play --task=hector [--load_run R --checkpoint K | --onnx actor.onnx]`

Headless counterpart of the reference's humanoid/scripts/play.py: same environment overrides (8 robots on a 5 x 5
trimesh tile map, noise level 0.5, pushes off; :47-64), the policy from the newest checkpoint of the experiment
(`train_cfg.runner.resume = True`, :74-76) or from an exported ONNX actor, policy export as TorchScript + ONNX
(:80-98), fixed command vx = 0.5 (:136-140), `play_steps` steps with the same per-step state log (:160-175) and the
reward summary (:183).  Rendering / video (:102-127,144-157) has no counterpart here; the traces are written to
`<out>/play_states.npz|csv` instead of a plot window.
"""
import os

import numpy as np

from isaac_amd import LEGGED_GYM_ROOT_DIR
from isaac_amd.envs import *  # noqa: F401,F403  (registers tasks)
from isaac_amd.utils import Logger, export_policy_as_jit, export_policy_as_onnx, get_args, task_registry

EXPORT_POLICY = True
FIX_COMMAND = True


def play(args):
    env_cfg, train_cfg = task_registry.get_cfgs(name=args.task)
    env_cfg.env.num_envs = 8
    env_cfg.terrain.mesh_type = "trimesh"
    env_cfg.terrain.num_rows = 5
    env_cfg.terrain.num_cols = 5
    env_cfg.terrain.curriculum = False
    env_cfg.terrain.max_init_terrain_level = 5
    env_cfg.noise.add_noise = True
    env_cfg.domain_rand.push_robots = False
    env_cfg.domain_rand.joint_angle_noise = 0.0
    env_cfg.noise.curriculum = False
    env_cfg.noise.noise_level = 0.5
    train_cfg.seed = 123145
    args.num_envs = None                                  # the override above wins, as in the reference

    env, _ = task_registry.make_env(name=args.task, args=args, env_cfg=env_cfg)
    obs = env.get_observations()
    train_cfg.runner.resume = args.onnx is None
    ppo_runner, train_cfg = task_registry.make_alg_runner(env=env, name=args.task, args=args, train_cfg=train_cfg,
                                                         log_root=None if args.onnx is not None else "default")
    if args.onnx is not None:
        ppo_runner.alg.actor_critic.load_actor_from_onnx(args.onnx)
        print("Loaded actor from:", args.onnx)
    policy = ppo_runner.get_inference_policy(device=env.device)

    out_dir = args.play_out or os.path.join(LEGGED_GYM_ROOT_DIR, "logs", train_cfg.runner.experiment_name, "exported")
    if EXPORT_POLICY:
        path = os.path.join(out_dir, "policies")
        export_policy_as_jit(ppo_runner.alg.actor_critic, path)
        print("Exported policy as jit script to: ", path)
        print("Exported policy as onnx to: ", export_policy_as_onnx(ppo_runner.alg.actor_critic, path))

    logger = Logger(env.dt)
    robot_index = -1      # which robot is used for logging
    joint_index = 9       # which joint is used for logging
    feet = env.feet_indices
    for _ in range(args.play_steps):
        actions = policy(obs)
        if FIX_COMMAND:
            c = env.commands
            c[:, 0], c[:, 1], c[:, 2], c[:, 3] = 0.5, 0.0, 0.0, 0.0
            env.commands = c
        obs, critic_obs, rews, dones, infos = env.step(actions)
        a = actions.numpy()
        cmd, lin, ang = env.commands, env.base_lin_vel, env.base_ang_vel
        logger.log_states({
            "dof_pos_target": float(a[robot_index, joint_index]) * env.cfg.control.action_scale,
            "dof_pos": float(env.dof_pos[robot_index, joint_index]),
            "dof_vel": float(env.dof_vel[robot_index, joint_index]),
            "dof_torque": float(env.torques[robot_index, joint_index]),
            "command_x": float(cmd[robot_index, 0]),
            "command_y": float(cmd[robot_index, 1]),
            "command_yaw": float(cmd[robot_index, 2]),
            "base_vel_x": float(lin[robot_index, 0]),
            "base_vel_y": float(lin[robot_index, 1]),
            "base_vel_z": float(lin[robot_index, 2]),
            "base_vel_yaw": float(ang[robot_index, 2]),
            "contact_forces_z": env.contact_forces[robot_index, feet, 2].copy(),
        })
        num_episodes = int(np.sum(dones.numpy()))
        if num_episodes > 0:
            info, _ = env.episode_stats()
            logger.log_rewards(info, num_episodes)
    logger.print_rewards()
    trace = logger.plot_states(os.path.join(out_dir, "play_states"))
    print("State traces written to:", trace)
    return logger


if __name__ == "__main__":
    play(get_args())
