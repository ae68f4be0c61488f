"""Worker of tests/test_parallel_cpu.py: one data-parallel rank on the CPU (gloo), driving the SAME
orchestration the GPU path uses (isaac_amd.parallel.TorchComm: broadcast, moments all-reduce, ONE flat
gradient+statistics all-reduce per optimiser step, 1/world scaling inside the step) with the numpy oracle
standing in for the HIP learner."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isaac_amd.parallel import TorchComm, shard_envs  # noqa: E402
from oracle.ppo import ActorCriticOracle, PPOOracle  # noqa: E402
from tests.ppo_inputs import rollout_inputs  # noqa: E402


def flat(gs):
    return np.concatenate([g.reshape(-1) for g in gs])


def run(out_path, total_envs=32, T=4, seed=3):
    comm = TorchComm("gloo")
    lo, hi = shard_envs(total_envs, comm.rank, comm.world_size)
    n = hi - lo
    # every rank builds DIFFERENT initial weights; broadcast must make rank 0's win
    ac = ActorCriticOracle.default_init(np.random.default_rng(100 + comm.rank), 24, 40, 4, (32, 16, 8), (32, 16, 8))
    sd = comm.broadcast_state(ac.state_dict())
    for k, v in ac.state_dict().items():
        v[...] = sd[k]
    alg = PPOOracle(ac, n, T, num_learning_epochs=1, num_mini_batches=2, learning_rate=1e-3)
    inp = rollout_inputs(seed, T, total_envs, 24, 40, 4)
    for t in range(T):
        alg.act(inp["obs"][t][lo:hi], inp["priv"][t][lo:hi], inp["eps"][t][lo:hi])
        alg.process_env_step(inp["rewards"][t][lo:hi], inp["dones"][t][lo:hi], inp["time_outs"][t][lo:hi])
    # GAE locally, advantage normalisation with GLOBAL moments (one 3-double all-reduce)
    alg.compute_returns(inp["priv"][T][lo:hi])
    raw = (alg.returns - alg.values).astype(np.float64)
    moments = np.array([raw.sum(), (raw * raw).sum(), raw.size], np.float64)
    comm.all_reduce_moments(moments.ctypes.data, None)
    mean = moments[0] / moments[2]
    std = np.sqrt((moments[1] - moments[2] * mean * mean) / (moments[2] - 1))
    alg.advantages = ((raw - mean) / (std + 1e-8)).astype(np.float32)
    # the flat buffer [grads | kl_sum, vloss_sum, sloss_sum, rows] lives in memory owned by the comm
    nparam = sum(p.size for p in ac.params())
    ptr = comm.alloc_grad_buffer(nparam + 4)
    buf = comm._grad.numpy()
    assert buf.ctypes.data == ptr
    perm = np.random.default_rng(9).permutation(n * T)
    mbs = n * T // 2
    hist = []
    for i in range(2):
        info, grads = alg.loss_and_grads(perm[i * mbs:(i + 1) * mbs])
        buf[:nparam] = flat(grads)
        buf[nparam:] = [info["kl"] * mbs, info["value"] * mbs, info["surrogate"] * mbs, mbs]
        comm.all_reduce_grads(ptr, nparam + 4, None)            # the ONE collective of this optimiser step
        kl = float(buf[nparam] / buf[nparam + 3])               # global KL -> identical LR decision on every rank
        alg.adapt_lr(kl)
        g = buf[:nparam] * np.float32(1.0 / comm.world_size)
        out, o = [], 0
        for p in ac.params():
            out.append(g[o:o + p.size].reshape(p.shape).copy())
            o += p.size
        alg.optimizer_step(out)
        hist.append((kl, alg.lr))
    np.savez(out_path.format(rank=comm.rank), params=flat(ac.params()), hist=np.array(hist), lo=lo, hi=hi,
             adv=alg.advantages)
    comm.barrier()


if __name__ == "__main__":
    run(sys.argv[1])
