"""`python -m isaac_amd.scripts.train --task=hector --headless --num_envs 4096`
(reference humanoid/scripts/train.py:36-43).  Under `python -m torch.distributed.run --nproc-per-node N` it
trains data-parallel: one process per GPU, envs sharded, one RCCL gradient all-reduce per optimiser step."""
from isaac_amd.envs import *  # noqa: F401,F403  (registers tasks)
from isaac_amd.parallel import init_comm
from isaac_amd.utils import get_args, task_registry


def train(args):
    comm = init_comm()
    env, env_cfg = task_registry.make_env(name=args.task, args=args, comm=comm)
    ppo_runner, train_cfg = task_registry.make_alg_runner(env=env, name=args.task, args=args, comm=comm)
    ppo_runner.learn(num_learning_iterations=train_cfg.runner.max_iterations, init_at_random_ep_len=True)
    comm.barrier()
    comm.close()


if __name__ == "__main__":
    train(get_args())
