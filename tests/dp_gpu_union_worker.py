"""Worker of tests/test_gpu_dp.py::test_two_ranks_equal_single_process_on_the_union: ONE logical batch of 512 robots, either
simulated and learned by a single process ("union") or split into env_range halves over two data-parallel ranks (gloo-staged
collectives on the one GPU of the test box).  Robots are keyed by their global id (hx_sim_cfg.env_id_offset), the action
noise by the global row (hx_ppo_set_row_base), and the permutations are constructed so that minibatch i of the union is the
union of the ranks' minibatches i -- so both runs see the same samples and must end with the same parameters up to the
order of fp32 sums (gradient of 2 x 1920 rows summed per rank then across ranks, vs 3840 rows in one reduction)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TOTAL, T, MB = 512, 15, 2


def main(mode, out_path):
    import torch
    from isaac_amd import capi
    from isaac_amd.algo.ppo import PPO, ActorCritic
    from isaac_amd.envs.configs import HectorCfg
    from isaac_amd.envs.hector_env import HectorFreeEnv
    from isaac_amd.parallel import Comm, init_comm
    from isaac_amd.utils.helpers import set_seed
    comm = init_comm() if mode == "rank" else Comm()
    capi.check(capi.lib().hx_set_device(0), "set_device")
    world, rank = comm.world_size, comm.rank
    n = TOTAL // world
    lo = rank * n
    cfg = HectorCfg()
    cfg.env.num_envs = TOTAL
    cfg.terrain.num_rows, cfg.terrain.num_cols = 4, 4
    cfg.seed = set_seed(5)                                       # the SAME seed everywhere: one logical batch
    env = HectorFreeEnv(cfg, sim_device="cuda:0", headless=True, env_range=None if world == 1 else (lo, lo + n))
    torch.manual_seed(7)
    ac = ActorCritic(env.num_obs, env.num_privileged_obs, env.num_actions, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[768, 256, 128])
    alg = PPO(ac, num_learning_epochs=2, num_mini_batches=MB, gamma=0.994, lam=0.9, entropy_coef=0.001, learning_rate=1e-4,
              schedule="adaptive", desired_kl=0.01, stream=env.stream, comm=comm)
    alg.init_storage(n, T, [env.num_obs], [env.num_privileged_obs], [env.num_actions], obs_ld=env.obs_ld, priv_ld=env.priv_ld)
    capi.check(capi.lib().hx_ppo_set_row_base(alg._h, lo), "hx_ppo_set_row_base")
    env.reset()
    half = TOTAL // 2
    loc = np.random.default_rng(9).permutation(half * T)          # the permutation a half-batch rank uses
    mbs = half * T // MB
    if world == 1:                                                # minibatch i = rank 0's minibatch i + rank 1's minibatch i
        parts = []
        for i in range(MB):
            t, e = loc[i * mbs:(i + 1) * mbs] // half, loc[i * mbs:(i + 1) * mbs] % half
            parts += [t * TOTAL + e, t * TOTAL + e + half]
        perm = np.concatenate(parts)
    else:
        perm = loc
    hist = []
    for it in range(2):
        alg.rollout([env], T)
        env.sync()
        alg.compute_returns(env.get_privileged_observations())
        adv = alg.buffer(capi.PPO_BUF_ADVANTAGES, (T, n)).numpy().copy()
        hist.append(alg.update(perm=perm.astype(np.int32)) + (alg.learning_rate,))
    sd = ac.state_dict()
    np.savez(out_path.format(rank=rank), params=np.concatenate([v.reshape(-1) for v in sd.values()]), adv=adv, hist=np.array(hist),
             actions=alg.buffer(capi.PPO_BUF_ACTIONS, (T, n, env.num_actions)).numpy())
    comm.barrier()
    alg.close()
    env.close()
    comm.close()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
