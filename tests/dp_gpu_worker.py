"""Worker of tests/test_gpu_dp.py: one data-parallel rank with the REAL simulator and learner (libhx.so) on the GPU.
Ranks share one GPU in the test, so the collectives run as "gloo-staged" (host copies of the device buffers); the
orchestration -- env shards with their own seeds, parameter broadcast, global advantage moments, one flat gradient +
statistics all-reduce per optimiser step, 1/world scaling in the Adam kernel -- is the production code."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out_path, iters):
    from isaac_amd import capi
    from isaac_amd.algo.on_policy_runner import OnPolicyRunner
    from isaac_amd.cfgtools import class_to_dict
    from isaac_amd.envs.configs import HectorCfg, HectorCfgPPO
    from isaac_amd.envs.hector_env import HectorFreeEnv
    from isaac_amd.parallel import init_comm
    from isaac_amd.utils.helpers import set_seed
    if os.environ.get("HX_DP_FORCE_RCCL") == "1":      # one rank, real RCCL: every collective of the N > 1 path runs (as identity)
        from isaac_amd.parallel import HxComm
        comm = HxComm(rank=0, world_size=1, local_rank=0)
        comm.force_collectives = True
    else:
        comm = init_comm()
    capi.check(capi.lib().hx_set_device(0), "set_device")
    env_cfg, train_cfg = HectorCfg(), HectorCfgPPO()
    env_cfg.env.num_envs = 256
    env_cfg.terrain.num_rows, env_cfg.terrain.num_cols = 4, 4
    env_cfg.seed = set_seed(train_cfg.seed + comm.rank)              # every rank its own robots and random streams
    env = HectorFreeEnv(env_cfg, sim_device="cuda:0", headless=True)
    runner = OnPolicyRunner(env, class_to_dict(train_cfg), log_dir=None, device="cuda:0", comm=comm)
    runner.learn(iters, init_at_random_ep_len=True)
    sd = runner.alg.actor_critic.state_dict()
    flat = np.concatenate([v.reshape(-1) for v in sd.values()])
    m, v, step = runner.alg.optimizer_state()
    first_obs = env.get_observations().numpy()[:4, -41:].copy()
    np.savez(out_path.format(rank=comm.rank), params=flat, m=m, v=v, step=step, lr=runner.alg.learning_rate, first_obs=first_obs)
    comm.barrier()
    runner.alg.close()
    env.close()


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))
